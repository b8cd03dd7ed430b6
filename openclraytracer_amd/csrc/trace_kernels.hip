// trace_kernels.hip -- the hot path: BVH traversal + ray/AABB + ray/triangle for gfx950.
//
// Semantics are those of bvh_generate_contact (Include/CLData/AccelerationStructs/BVH.h:344-408):
// a child is entered iff (AABBIntersect > 0 || origin inside the box) (BVH.h:366-369, AABB.h:126-142,
// 178-184), there is NO closest-hit pruning, and the closest hit is kept with a strict `<` in an
// A-first depth-first walk (BVH.h:371-392).  Hence
//      result(ray) = lexicographic min of (t, DFS rank of the leaf) over every REACHED leaf with t > 0,
// and because the radix tree is built over the sorted (morton, leaf) array, the DFS rank of a leaf is its
// position in that array (child A covers [first,split], child B [split+1,last], BVH.h:299-305; checked
// against an explicit DFS of the oracle's tree in tests).  The set of reached leaves does not depend on
// the order in which sub-trees are walked, so ANY traversal order and ANY split of one ray's sub-trees
// over lanes gives the reference's contact bit for bit, as long as ties are broken by that position.
// The arithmetic itself is restated operation by operation (device_math.h, no FMA contraction).
//
// For the same reason the hierarchy itself is free: a leaf whose box passes the child test has only ancestors that pass
// it in ANY hierarchy of nested boxes over these leaves (monotone rounding), so rays may walk the
// "search tree" -- a second hierarchy over the same leaves in properly normalised Morton order, appended to the reference
// tree's records -- unless a slab product could be 0 * inf (ray_root).
//
// Machine mapping (MI355X, wave64, 256 CUs).  A frame is one first-phase kernel plus, when work was handed over,
// k_trace_long (+ k_resolve_split); which first-phase kernel and which budget is calibrated per BVH on the first frame
// (rtmi_capi.hip, trace_primary_tuned).  The first-phase kernels run one wave per 64-ray block (8x8 pixels of a primary
// frame) with the hardware dispatcher as the work queue.
//  k_trace, one ray per lane:
//   * the per-lane traversal stack lives in LDS ([level][lane]: one bank per lane, no conflicts);
//   * one inner-node visit = one 64-byte record holding both child boxes and both child ids (TravNode)
//     instead of a 48-byte node + two dependent 32-byte box fetches (BVH.h:359-364) -- or, on search-tree records, one
//     128-byte record holding the node's up to four grandchildren (QuadNode): half the dependent fetches;
//   * leaves read an SoA triangle table (v0, e1, e2 as float4) instead of chasing ModelHeader /
//     MeshHeader.dataSize chains and ushort indices (SceneBufferParser.h:175-217); edges and the
//     geometric normal are ray independent and precomputed with the same operations;
//   * leaf tests are postponed: a lane parks one reached leaf and keeps walking inner nodes; the wave
//     runs the triangle code only when a lane needs its slot again or is about to finish;
//   * a ray that is still walking after `defer_budget` steps, or whose stack is full, is DEFERRED: its state (pending
//     sub-trees, best hit so far) goes to a compact list.  A few rays visit thousands of nodes while most
//     visit a handful; walked by one lane each they would set the frame time.
//  k_trace_packet, one 8x8-pixel block per wave: the node is wave-uniform, its record comes through the scalar path
//     and stays in registers, both children's records are requested before the box tests; (node, lane mask) pairs wait
//     on an LDS stack; sub-trees only a handful of lanes still want, or everything left once a budget is spent, go to
//     the deferred list as per-ray partial records whose results meet in a 64-bit atomicMin on (t, position).
//  k_trace_packet4, the kernel most frames run on: the same walk on the four-wide records of the search tree (two levels per
//     step), on per-camera copies of those records and of the triangles (k_prepare_camera, k_prepare_triangles): box tests
//     of 10 vector instructions whose results are lane masks in scalar registers, the stack in lanes of vector registers,
//     optionally (SHARE) the waiting sub-trees of long walks published as tasks for other waves, optionally (FUSE, frame
//     graphs) the colour written next to the contact.  Described where it stands, below.
//  k_trace_long, one, two or four deferred records per WAVE: the 64 lanes walk the rays' sub-trees together out of an LDS
//     node pool (64 nodes per step, children appended with ballot/prefix-popcount compaction, spare lanes running ahead
//     along duplicate-code chains, leaves collected in a second pool and tested 64 at a time); hits meet in a 64-bit
//     (t, position) key per ray through an LDS atomic min.
//  k_prepare_camera / k_prepare_triangles, per camera position: "camera inside child box" bits of the binary records, the
//     camera-relative copies of the four-wide records (one per direction octant) and of the triangles, the camera-on-a-plane bits.
#include "rtmi_internal.h"
#include "device_math.h"
#include "shade_math.h"
#include <vector>
#include <cstdio>
#include <cstdlib>

// This file is compiled TWICE into librtmi.so (csrc/Makefile): as it stands -- dot() / cross() of the reference's HOST branch, separate
// multiplies and adds (CLPortability.h:266-278): the arithmetic of RT_MODE_FIXED / LITERAL_ZERO / FAST -- and with -DRTMI_DEVICE_ARITH=1 --
// dot() / cross() as the OpenCL C built-ins the reference's DEVICE branch calls (CLPortability.h:41-45), which ROCm's device library fuses
// (device_math.h) -- for BVHs created with RT_MODE_DEVICE_ARITH.  The second compilation renames every kernel and entry point below
// (suffix _devarith); the first one's entry points hand a launch over when TraceLaunch::device_arith is set.
#if defined(RTMI_DEVICE_ARITH) && RTMI_DEVICE_ARITH
#define k_prepare_camera k_prepare_camera_devarith
#define k_device_clock k_device_clock_devarith
#define k_trace k_trace_devarith
#define k_trace_packet k_trace_packet_devarith
#define k_trace_packet4 k_trace_packet4_devarith
#define k_trace_long k_trace_long_devarith
#define k_resolve_split k_resolve_split_devarith
#define k_block_order k_block_order_devarith
#define launch_trace_primary launch_trace_primary_devarith
#define launch_trace_rays launch_trace_rays_devarith
#define launch_prepare_camera launch_prepare_camera_devarith
#define launch_device_clock launch_device_clock_devarith
#define launch_block_order launch_block_order_devarith
#define rt_preload_trace rt_preload_trace_devarith
#define trace_lds_stack_entries trace_lds_stack_entries_devarith
#define trace_ctrl_bytes trace_ctrl_bytes_devarith
#define trace_ctrl_error_offset trace_ctrl_error_offset_devarith
#define trace_ctrl_expired_offset trace_ctrl_expired_offset_devarith
#define trace_ctrl_counter_offsets trace_ctrl_counter_offsets_devarith
#define trace_long_record_bytes trace_long_record_bytes_devarith
#define trace_task_bytes trace_task_bytes_devarith
#define trace_packet_stack_entries trace_packet_stack_entries_devarith
#define RTMI_TRACE_HANDS_OVER 0
#else
#define RTMI_TRACE_HANDS_OVER 1
#endif

#ifndef TRACE_BLOCK
#define TRACE_BLOCK 256u
#endif
#define TRACE_WAVES (TRACE_BLOCK / 64u)
#define LDS_STACK 24u            // stack entries per lane in k_trace (observed need at 1080p: <= 18)
#define NO_LEAF 0xFFFFFFFEu
#define DEFER_BUDGET_DEFAULT 1024u  // k_trace iterations after which a ray moves to the deferred list (default when nothing was calibrated)
#define LONG_REC_WORDS 32u       // 128-byte record per deferred ray
#define POOL_INNER 1024u         // phase-1 LDS pool of pending inner nodes (per wave)
#define POOL_LEAF 256u           // phase-1 LDS pool of pending leaves (per wave)

typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

uint32_t trace_lds_stack_entries() { return LDS_STACK; }

// control block; the counters are zeroed before every launch, every word on its own 128-byte line.  The error word is
// STICKY: kernels OR bit `code` into it, only the host clears it (check_trace_error), so an invalid frame anywhere in a batch
// of launches is still seen by the next synchronising call.
#define WORK_SHARDS 16u
struct TraceCtrl {
    struct { uint32_t next; uint32_t pad[31]; } lshard[WORK_SHARDS];  // wave-per-ray kernel: next batch of deferred rays per shard
    uint32_t long_count;  uint32_t p1[31];   // deferred rays
    uint32_t split_count; uint32_t p3[31];   // rays whose sub-trees were handed to phase 1 by the packet kernel
    uint32_t spill_taken; uint32_t p4[31];   // packet kernels: overflow areas taken in this launch (diagnostic)
    uint32_t long_steps;  uint32_t long_steps_max; uint32_t long_nodes; uint32_t p5[29];   // wave-per-ray kernel: pool steps, all records / longest record (diagnostic)
    uint32_t split_tiles; uint32_t shared_tasks; uint32_t expired; uint32_t p9[29];   // k_trace_packet4 with shared tasks: blocks whose contacts come out of the key buffer (entries of
                                             // split_list); tasks published (diagnostic, written by the helper waves at the end)
    uint32_t error;       uint32_t p2[31];   // bit 1: long-ray list full on a forced deferral, 2: phase-1 pool overflow, 3: packet stack
                                             // beyond its spill area, 4: bad deferred record, 5: no free spill area, 6: a ray was handed over
                                             // in a frame captured without the wave-per-ray kernel, 7: a ray of a direction octant whose record
                                             // copy was not prepared.  NOT zeroed per launch.
};
static_assert(sizeof(TraceCtrl) == (WORK_SHARDS + 6) * 128, "TraceCtrl layout");
static_assert(offsetof(TraceCtrl, error) == (WORK_SHARDS + 5) * 128, "the error word follows everything a launch clears");
#define TRACE_FAIL(word, code) __hip_atomic_fetch_or((word), 1u << (code), RLX_AGENT)
size_t trace_ctrl_bytes() { return sizeof(TraceCtrl); }
size_t trace_ctrl_error_offset() { return offsetof(TraceCtrl, error); }
size_t trace_ctrl_expired_offset() { return offsetof(TraceCtrl, expired); }      // calibration launches: a wave stopped at the deadline
void trace_ctrl_counter_offsets(size_t out[RT_TRACE_COUNTERS]) { out[6] = offsetof(TraceCtrl, split_tiles); out[7] = offsetof(TraceCtrl, shared_tasks);
                                                  out[0] = offsetof(TraceCtrl, long_count); out[1] = offsetof(TraceCtrl, split_count); out[2] = offsetof(TraceCtrl, spill_taken);
                                                  out[3] = offsetof(TraceCtrl, long_steps); out[4] = offsetof(TraceCtrl, long_steps_max); out[5] = offsetof(TraceCtrl, long_nodes); }
size_t trace_long_record_bytes() { return LONG_REC_WORDS * sizeof(uint32_t); }

// record of a deferred ray (LONG_REC_WORDS words)
enum { LR_WORK = 0, LR_CUR = 1, LR_SP = 2, LR_BEST_LEAF = 3, LR_BEST_T = 4, LR_EXTRA = 5, LR_PENDING = 6, LR_PARTIAL = 7, LR_STACK = 8 };
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull
typedef __attribute__((address_space(1))) unsigned long long gu64;

// Sub-tree tasks shared between the waves of k_trace_packet4.  An operation on one address costs ~15 ns at the memory side
// whoever issues it (32 400 waves bumping one counter: 0.5 ms), so the queue is TASK_SHARDS queues, each with its own line:
// `published` (a publisher's fetch-add hands it a slot; slots at or beyond the capacity do not exist, the entry stays with its
// publisher), `taken` (a taker's compare-and-swap, only ever up to min(published, capacity)), the `finished` word the helpers
// leave on, and the shard's counters of finished block waves and tasks.  A publisher spreads its entries over the shards
// (lane i: shard home + i), a finished block wave looks at one of them, a helper at two per round.
// Entry = (block's first work item + 1 [0 = not yet written], node, mask low, mask high), written and read as 16 bytes.
#define TASK_SHARDS 256u
#define P4_DBG_JOBS 48u
// steps between two sharings of a wave once the halving interval has come down (round 3, tight tree: 8 instead of 32 -- Tenryuu shared
// after 32 steps 0.138 -> 0.123 ms, ClumsyDragon after 16 steps 0.158 -> 0.126; ManFigure unchanged)
#ifndef SHARE_MIN_STEPS
#define SHARE_MIN_STEPS 8u
#endif
#ifndef SHARE_BOTTOM_HALF
#define SHARE_BOTTOM_HALF 1
#endif
struct TaskShard { uint32_t taken; uint32_t published; uint32_t finished; uint32_t pad0; uint32_t waves_done; uint32_t tasks_done; uint32_t pad[26]; };
static_assert(sizeof(TaskShard) == 128, "one line per shard");
size_t trace_task_bytes(uint32_t cap_per_shard) { return TASK_SHARDS * (sizeof(TaskShard) + (size_t)cap_per_shard * 16u); }

struct TraceArgs {
    const TravNode* __restrict__ trav;
    uint32_t quad_octant_records;               // primary rays: records per copy when the per-octant copies exist (copy 1 + octant), else 0
    const QuadNode* __restrict__ quad;          // four-wide records of the same nodes, may be null: what the kernels read -- for primary
                                                // rays the camera-relative copy (boxes minus the camera position, inside marks)
    uint32_t search_root;                       // primary rays: root record of the search tree (same leaves), 0 = none
    const uint32_t* __restrict__ tree_select;   // primary rays: bit k = the camera's coordinate k equals a box plane
    uint32_t plane_bits_known;                  // 0x80000000 | those bits when the host knows them (movable frame graphs): tree_select is not read
    uint32_t octant_mask;                       // four-wide kernel on per-octant records: bit o = the copy of octant o is valid for this camera
    uint32_t no_unsafe_rays;                    // the launcher left the wave-per-ray kernel out: a ray handed over would be lost (error bit 6)
    uint32_t cleared_by_preparation;            // host only: the preparation kernel in front of this launch zeroed the control block and the task queues
    const uint32_t* __restrict__ block_order;   // four-wide kernel: wave -> 8x8 block (null: identity), the most expensive blocks of an earlier frame first
    uint32_t* block_cost;                       // ... steps per block of this frame, tasks included (null: not counted)
    unsigned long long cal_deadline;            // calibration launches (CAL = 1) only: device time (s_memrealtime) after which waves stop walking, 0 = none
    const float4* __restrict__ v0;
    const float4* __restrict__ e1;
    const float4* __restrict__ e2;
    const float4* __restrict__ nrm;
    const TriRec* __restrict__ trirec;     // packed triangle records (packet kernels): vertex 0, edges, rank behind one address
    const TriCam* __restrict__ tricam;     // primary rays: the camera-relative triangle records (k_trace_packet4)
    const uint32_t* __restrict__ pos;      // leaf -> position in the sorted (morton, leaf) array
    TraceCtrl* ctrl;
    uint32_t* long_list;                   // long_cap records
    uint32_t long_cap;
    uint32_t defer_budget;
    uint32_t total_work;                   // work items (multiple of 4096 for primary frames)
    uint32_t n_waves;                      // waves in the grid
    uint32_t n_records;                    // inner records in `trav`: leaves - 1 per tree
    uint32_t n_leaves;
    unsigned long long* keys;              // (t bits << 32 | position) per work item, used by split rays only
    uint32_t* split_list;                  // work items of split rays (long_cap entries)
    const uint2* __restrict__ pairs;       // position -> leaf
    uint32_t* packet_spill;                // packet kernels: overflow areas of the (node, mask) stack, handed out on demand;
    uint32_t packet_spill_entries;         // the first packet_spill_slots / 32 words are the bitmap of areas in use
    uint32_t packet_spill_slots;
    uint32_t long_batch;                   // experiment knob: rays per wave in the wave-per-ray kernel (0 = by count)
    uint32_t diagnostics;                  // count the wave-per-ray kernel's pool steps (perturbs its timing)
    TaskShard* task_shards;                // k_trace_packet4 with shared tasks: TASK_SHARDS queues, then their entries
    uint32_t task_cap;                     // entries per queue
    // fused frame: shade a contact where it is stored (k_trace_packet4 only; null = contacts only)
    float4* fused_rgba;
    uint32_t fused_plane_stride;
    ShadeScene fused_scene;
#if RTMI_P4_VARIANT == 3
    unsigned long long* dbg;               // per wave P4_DBG_JOBS x (start, end, steps << 1 | is_task) on the 100 MHz clock
#endif
    // primary rays
    rt_camera cam;
    uint32_t rank, nranks, tiles_x;
    const uint2* __restrict__ tile_xy;          // primary frames: pixel origin (x0, y0) of this rank's tile `ordinal` (null: computed per lane)
    uint32_t compact;
    // ray buffer
    const rt_ray* __restrict__ rays;
    rt_contact* __restrict__ out;
};

struct RayState {
    f3 o, d, inv;
    float best_t;
    uint32_t best_leaf;
    uint32_t best_pos;       // packet kernels: depth-first rank of best_leaf (the tie rule without a load)
    uint32_t work;           // work item (ray) index
    uint32_t out_index;      // where the contact is stored
    uint32_t pixel;          // Contact.pixelIndex
};

// one child of an inner node: AABBIntersect (AABB.h:126-142) folded with the validity rule of
// BVH.h:368-369.  `tmin*flag > 0` <=> `flag && tmin > 0` (x*1.0f == x; x*0.0f is +-0 or NaN).
// Branch free on purpose (bit-wise & and |): both children are always evaluated.
// Primary rays share one origin (the camera position), so "origin inside the box" does not depend on the ray:
// k_prepare_camera evaluates it once per node and camera position and the primary kernels read the bit.
template <int PRIMARY>
__device__ __forceinline__ bool child_valid(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                            const RayState& r, bool sx, bool sy, bool sz, bool inside_marked)
{
    float tmin  = ((sx ? hix : lox) - r.o.x) * r.inv.x;
    float tmax  = ((sx ? lox : hix) - r.o.x) * r.inv.x;
    float tymin = ((sy ? hiy : loy) - r.o.y) * r.inv.y;
    float tymax = ((sy ? loy : hiy) - r.o.y) * r.inv.y;
    bool flag = !((tmin > tymax) | (tymin > tmax));
    tmin = RMAX(tmin, tymin);
    tmax = RMIN(tmax, tymax);
    float tzmin = ((sz ? hiz : loz) - r.o.z) * r.inv.z;
    float tzmax = ((sz ? loz : hiz) - r.o.z) * r.inv.z;
    flag = flag & !((tmin > tzmax) | (tzmin > tmax));
    tmin = RMAX(tmin, tzmin);
    bool inside = PRIMARY ? inside_marked
                          : ((lox <= r.o.x) & (r.o.x <= hix) & (loy <= r.o.y) & (r.o.y <= hiy) & (loz <= r.o.z) & (r.o.z <= hiz));
    return (flag & (tmin > 0.0f)) | inside;
}

// The same decision for rays whose slab products cannot be NaN -- every ray the four-wide records are walked with: they live
// on the search tree only, and ray_root keeps rays with a possible 0 * inf off it (a ray with a NaN direction can pass, but no
// triangle test of such a ray succeeds whatever it visits).  Without NaN the reference's staged test is value-equivalent to the
// textbook slab test: per axis the two products a = (lo - o) * inv, b = (hi - o) * inv are ordered by the sign of inv, so the
// sign-selected near / far planes are min(a, b) / max(a, b); "no near exceeds another axis' far" (AABB.h:133-139) together with
// near_k <= far_k (lo <= hi, monotone rounding) is max3(near) <= min3(far); tmin is max3(near).  Signed zeros compare equal.
// This form is pure VALU -- v_min / v_max / v_max3 / v_min3, no compare-into-SGPR + select round trips between the vector
// and the scalar unit, which is what a wave step of the packet kernel waited for.
// (the per-lane kernels evaluate it with the planes selected by sign: child_valid_signed below)

// ... and on the camera-relative records of primary rays (k_prepare_camera): d* = plane - origin, subtracted once per camera
__device__ __forceinline__ bool child_valid_delta(float dlox, float dloy, float dloz, float dhix, float dhiy, float dhiz, const RayState& r,
                                                  float threshold)                 // 0, or -inf with the camera inside the box
{
    const float ax = dlox * r.inv.x, bx = dhix * r.inv.x;
    const float ay = dloy * r.inv.y, by = dhiy * r.inv.y;
    const float az = dloz * r.inv.z, bz = dhiz * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return (tnear <= tfar) & (tnear > threshold);
}

// The same two tests with the near / far planes SELECTED by the direction's sign instead of min / max of the two products (the same
// values whenever neither product is NaN, and the same number of vector instructions), for the lane-per-ray and wave-per-ray kernels
// on search-tree records: a 0 * inf on the x axis then stays a NaN, which v_max_f32 / v_min_f32 ignore -- the extended-real form of the
// test that DESIGN.md section 2 ("the first axis") shows to be the reference's for such rays, and the form the four-wide kernel's
// per-octant records evaluate (child_mask_octant).  min / max of the products would turn that NaN into the wrong infinity.  So since
// round 3 a ray of these kernels leaves the search tree only for a possible 0 * inf on y or z (ray_root with the x bit cleared):
// ClumsyDragon's 1 080 centre-column rays (camera x = 0 is a plane of that model, direction.x = 0) no longer walk the reference tree,
// whose boxes of the model's whole negative half reach up to the plane x = 0 (FLT_MIN) and all contain such a ray's x.
__device__ __forceinline__ bool child_valid_signed(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayState& r,
                                                   bool sx, bool sy, bool sz, bool inside_marked)
{
    const float nx = ((sx ? hix : lox) - r.o.x) * r.inv.x, fx = ((sx ? lox : hix) - r.o.x) * r.inv.x;
    const float ny = ((sy ? hiy : loy) - r.o.y) * r.inv.y, fy = ((sy ? loy : hiy) - r.o.y) * r.inv.y;
    const float nz = ((sz ? hiz : loz) - r.o.z) * r.inv.z, fz = ((sz ? loz : hiz) - r.o.z) * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    const float tfar = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return ((tnear <= tfar) & (tnear > 0.0f)) | inside_marked;
}
__device__ __forceinline__ bool child_valid_delta_signed(float dlox, float dloy, float dloz, float dhix, float dhiy, float dhiz, const RayState& r,
                                                         bool sx, bool sy, bool sz, float threshold)
{
    const float nx = (sx ? dhix : dlox) * r.inv.x, fx = (sx ? dlox : dhix) * r.inv.x;
    const float ny = (sy ? dhiy : dloy) * r.inv.y, fy = (sy ? dloy : dhiy) * r.inv.y;
    const float nz = (sz ? dhiz : dloz) * r.inv.z, fz = (sz ? dloz : dhiz) * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    const float tfar = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return (tnear <= tfar) & (tnear > threshold);
}

// A ray of a RAY BUFFER on search-tree records (ray_root<0>: finite origin of moderate size, every 1 / direction finite and at least
// ~1 in magnitude -- no NaN, no product that underflows to zero): "origin inside the box" is then tnear <= 0 <= tfar, and the
// reference's (flag && tmin > 0) || inside is tnear <= tfar && tfar >= 0 (derivation at child_mask_general).
__device__ __forceinline__ bool child_valid_general_signed(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayState& r,
                                                           bool sx, bool sy, bool sz)
{
    const float nx = ((sx ? hix : lox) - r.o.x) * r.inv.x, fx = ((sx ? lox : hix) - r.o.x) * r.inv.x;
    const float ny = ((sy ? hiy : loy) - r.o.y) * r.inv.y, fy = ((sy ? loy : hiy) - r.o.y) * r.inv.y;
    const float nz = ((sz ? hiz : loz) - r.o.z) * r.inv.z, fz = ((sz ? loz : hiz) - r.o.z) * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    const float tfar = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return (tnear <= tfar) & (tfar >= 0.0f);
}

// ... as the mask of the wave's lanes that enter the box.  The two compares write their lane masks straight into scalar
// registers and the rest is scalar arithmetic on 64-bit masks; as a per-lane bool the compiler turns every `__ballot` of it into
// a select and a compare of its own (two vector instructions per slot and step).
__device__ __forceinline__ uint64_t child_mask_delta(float dlox, float dloy, float dloz, float dhix, float dhiy, float dhiz, const RayState& r,
                                                     float threshold)
{
    const float ax = dlox * r.inv.x, bx = dhix * r.inv.x;
    const float ay = dloy * r.inv.y, by = dhiy * r.inv.y;
    const float az = dloz * r.inv.z, bz = dhiz * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return __builtin_amdgcn_ballot_w64(tnear <= tfar) & __builtin_amdgcn_ballot_w64(tnear > threshold);
}

// ... and for the rays of a RAY BUFFER walked as a packet (k_trace_packet4<0, ..>): every lane has an origin of its own, so the
// subtraction stays per lane, and so does "origin inside the box" (isPointInside, AABB.h:178-184) -- which the slab values already
// hold.  Only rays that ray_root lets onto the search tree come here: finite origins of moderate size, every 1 / direction finite and
// at least ~1 in magnitude.  For such a ray no slab product is NaN, and none underflows to zero: the sign of an fp32 difference of
// finite numbers is exact, and a non-zero difference times a factor of magnitude >= 0.99 rounds to a non-zero number.  Hence
//     origin inside (lo <= o <= hi on every axis)  <=>  every near product <= 0 and every far product >= 0  <=>  tnear <= 0 <= tfar,
// and the reference's decision  (flag && tmin > 0) || inside  (BVH.h:366-369) is
//     tnear <= tfar  &&  (tnear > 0 || (tnear <= 0 && tfar >= 0))   ==   tnear <= tfar && tfar >= 0
// -- the textbook test, here derived from the reference's.  (The equivalence needs lo <= hi; an unused slot's box is NaN on every
// plane, k_finish: all its products are NaN, v_min / v_max return NaN for two NaN operands, and both compares are false.)
__device__ __forceinline__ uint64_t child_mask_general(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayState& r)
{
    const float ax = (lox - r.o.x) * r.inv.x, bx = (hix - r.o.x) * r.inv.x;
    const float ay = (loy - r.o.y) * r.inv.y, by = (hiy - r.o.y) * r.inv.y;
    const float az = (loz - r.o.z) * r.inv.z, bz = (hiz - r.o.z) * r.inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return __builtin_amdgcn_ballot_w64(tnear <= tfar) & __builtin_amdgcn_ballot_w64(tfar >= 0.0f);
}

// ... and on a per-octant copy of those records, for a wave whose rays all have the copy's direction octant: the near planes
// stand where the low corner stood and the far planes where the high corner stood (k_prepare_camera), so the products are the
// near and far parameters themselves
__device__ __forceinline__ uint64_t child_mask_octant(float nx, float ny, float nz, float fx, float fy, float fz, const RayState& r, float threshold)
{
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(nx * r.inv.x, ny * r.inv.y), nz * r.inv.z);
    const float tfar = __builtin_fminf(__builtin_fminf(fx * r.inv.x, fy * r.inv.y), fz * r.inv.z);
    return __builtin_amdgcn_ballot_w64(tnear <= tfar) & __builtin_amdgcn_ballot_w64(tnear > threshold);
}

// a child of a BINARY record: the pure-VALU, sign-selected form on search-tree records (a NaN only on x there, and ignored: see above),
// the reference's staged form on reference-tree records (where rays with a possible 0 * inf on y or z walk)
template <int PRIMARY>
__device__ __forceinline__ bool child_valid_on(bool search_tree_record, float lox, float loy, float loz, float hix, float hiy, float hiz,
                                               const RayState& r, bool sx, bool sy, bool sz, bool inside_marked)
{
    if (search_tree_record) {
        if (!PRIMARY) return child_valid_general_signed(lox, loy, loz, hix, hiy, hiz, r, sx, sy, sz);
        return child_valid_signed(lox, loy, loz, hix, hiy, hiz, r, sx, sy, sz, inside_marked);
    }
    return child_valid<PRIMARY>(lox, loy, loz, hix, hiy, hiz, r, sx, sy, sz, inside_marked);
}

// Per camera position: (1) isPointInside (AABB.h:178-184) of the camera position for both children of every inner
// record of both trees, bits 0 / 1 of pad0; (2) whether the camera lies exactly on a box plane.  Only then can a slab
// product be 0 * inf = NaN (a ray with a zero direction component), which is the one case the equivalence argument
// of the search tree excludes: bit k of *select says "the camera's coordinate k is a box plane", and
// rays whose 1/direction is not finite on such an axis start at the reference tree's root instead (ray_root).
// Every box of either tree appears as a child box of the reference records, or is a union of such boxes whose planes
// are planes of leaf boxes or the FLT_MIN the reference starts its maxima from (BVH.h:310).
__device__ __forceinline__ uint32_t inside_bits(const TravNode& t, float ox, float oy, float oz)
{
    bool ia = (t.a_lo[0] <= ox) & (ox <= t.a_hi[0]) & (t.a_lo[1] <= oy) & (oy <= t.a_hi[1]) & (t.a_lo[2] <= oz) & (oz <= t.a_hi[2]);
    bool ib = (t.b_lo[0] <= ox) & (ox <= t.b_hi[0]) & (t.b_lo[1] <= oy) & (oy <= t.b_hi[1]) & (t.b_lo[2] <= oz) & (oz <= t.b_hi[2]);
    return (ia ? 1u : 0u) | (ib ? 2u : 0u);
}
// One kernel per camera position, three kinds of blocks (one launch, one graph node):
//   quads     eight lanes per four-wide record of the tree that is walked with them (the search tree), one 16-byte part each: loads
//             and stores are contiguous 128-byte lines per record, the partner's half of a box comes through a lane exchange;
//   marks     one thread per binary record of both trees: inside bits, camera-on-a-plane bits;
//   triangles one thread per triangle: the camera-relative record.
// `parts` (RT_PREP_*) says what is written: a strategy prepares what it reads -- the four-wide packet kernel the copies of the
// direction octants its camera can produce (the demo camera: four of eight) and the triangles, the lane-per-ray kernel the
// general copy, the wave-per-ray and binary packet kernels the marks -- since a moving camera pays this per frame
// (OpenCLRayTracerDemo.cpp:275-276 updates the camera before every RTFrame).
struct PrepareArgs {
    TravNode* trav; uint32_t n_records, n_reference;
    const QuadNode* quad; QuadNode* quad_cam; uint32_t quad_first, quad_count; unsigned long long copy_stride;   // copy c of record i: quad_cam[c * copy_stride + i]
    const TriRec* trirec; TriCam* tricam; uint32_t n_tris;
    float ox, oy, oz;
    uint32_t* select;
    uint32_t parts;
    uint32_t blocks_quads, blocks_marks;       // block ranges: quads, then marks, then triangles
    uint32_t* clear_a; uint32_t clear_a_words; // words the trace launch behind this kernel wants zeroed (its control block, its task queues): an eager
    uint32_t* clear_b; uint32_t clear_b_words; // frame's fills, done here on the way (a fill is a launch of its own, ~4 us)
};
__device__ __forceinline__ float4 shfl_xor4(const float4& v, int m)
{
    return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}
__global__ __launch_bounds__(256) void k_prepare_camera(PrepareArgs a)
{
    {
        const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
        for (uint32_t i = first; i < a.clear_a_words; i += stride) a.clear_a[i] = 0u;
        for (uint32_t i = first; i < a.clear_b_words; i += stride) a.clear_b[i] = 0u;
    }
    const float ox = a.ox, oy = a.oy, oz = a.oz;
    if (blockIdx.x < a.blocks_quads) {
        // the four-wide record relative to this camera: every primary ray has the camera position as origin, so the slab
        // test's (plane - origin) is the same subtraction for all of them -- done once here, with the same fp32 operation, and
        // the kernels multiply by 1 / direction directly.  Part j of a record: q[j] = (low corner of slot j / 2, .) for even j,
        // (high corner, .) for odd j; .w of parts 0-3 are the slots' ids, .w of parts 4-7 their thresholds (below).
        const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
        const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, base = lane & ~7u;
        uint32_t rec = gid >> 3;
        const bool live = rec < a.quad_count;
        rec = a.quad_first + (live ? rec : 0u);
        const float4 v = a.quad[rec].q[j];
        const float4 p = shfl_xor4(v, 1);                                 // the other corner of my slot
        const float4 lo = (j & 1u) ? p : v, hi = (j & 1u) ? v : p;
        const uint32_t id = __float_as_uint(__shfl(v.w, (int)(base + (j >> 1)), 64));
        // an unused slot: a box no ray enters (near = +-FLT_MAX / direction is never both positive and below far), so that the
        // kernels need not look at the id before the test
        const bool unused = id == RT_SENTINEL;
        // camera inside the box: the reference takes the child whatever the slab test says (AABB.h: flag | inside).  With the
        // origin inside, near <= 0 <= far holds per axis in floating point as well (the signs of the products; a 0 * inf is
        // excluded by ray_root, or ignored by max3 / min3 on the x axis), so "near <= far" is true by itself and the child is taken
        // iff "near > 0" is replaced by "near > -inf", true for every ray with one finite 1 / direction: the slot's THRESHOLD,
        // a value in the record instead of a mark the kernels would have to test and combine per slot and step
        const bool inside = !unused & (lo.x <= ox) & (ox <= hi.x) & (lo.y <= oy) & (oy <= hi.y) & (lo.z <= oz) & (oz <= hi.z);
        const float thr_slot = inside ? -__builtin_inff() : 0.0f;          // of slot j / 2 (both of its lanes hold it)
        const float thr_part = __shfl(thr_slot, (int)(base + 2u * (j & 3u)), 64);   // part 4 + k carries slot k's
        float4 d;
        if (unused) { const float e = (j & 1u) ? -FLT_MAX : FLT_MAX; d.x = e; d.y = e; d.z = e; }
        else { d.x = v.x - ox; d.y = v.y - oy; d.z = v.z - oz; }
        d.w = j < 4u ? v.w : thr_part;
        const float4 dp = shfl_xor4(d, 1);
        if (!live) return;
        if (a.parts & RT_PREP_GENERAL) a.quad_cam[rec].q[j] = d;
        // One copy per direction octant (bit k of the octant: 1 / direction negative on axis k) with each slot's NEAR planes
        // where the low corner stands and its FAR planes where the high corner stands.  For a ray of that octant the products
        // (plane - origin) * (1 / direction) are ordered by the sign of 1 / direction on every axis (lo <= hi, monotone
        // rounding), so min(a, b) / max(a, b) of the general test ARE these products: a wave whose rays share an octant
        // multiplies and takes max3 / min3, six vector instructions per box less, same values.
#pragma unroll
        for (uint32_t o = 0; o < 8u; o++) {
            if (!(a.parts & RT_PREP_OCTANT(o))) continue;
            float4 w = d;
            if (o & 1u) w.x = dp.x;
            if (o & 2u) w.y = dp.y;
            if (o & 4u) w.z = dp.z;
            a.quad_cam[(unsigned long long)(1u + o) * a.copy_stride + rec].q[j] = w;
        }
        return;
    }
    if (blockIdx.x < a.blocks_quads + a.blocks_marks) {
        const uint32_t i = (blockIdx.x - a.blocks_quads) * blockDim.x + threadIdx.x;
        if (i >= a.n_records) return;
        const TravNode t = a.trav[i];
        a.trav[i].pad0 = inside_bits(t, ox, oy, oz);
        if (i < a.n_reference && a.n_records > a.n_reference) {       // plane test over the reference records, when there is a search tree
            const float o[3] = { ox, oy, oz };
            uint32_t bits = 0;
            for (int k = 0; k < 3; k++)
                if ((o[k] == t.a_lo[k]) | (o[k] == t.a_hi[k]) | (o[k] == t.b_lo[k]) | (o[k] == t.b_hi[k]) | (o[k] == FLT_MIN)) bits |= 1u << k;
            if (bits) __hip_atomic_fetch_or(a.select, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // the camera-relative triangle records (rtmi_internal.h, TriCam): the operations of leaf_test that involve the origin and the
    // triangle only, in its order
    const uint32_t i = (blockIdx.x - a.blocks_quads - a.blocks_marks) * blockDim.x + threadIdx.x;
    if (i >= a.n_tris) return;
    const TriRec t = a.trirec[i];
    const f3 vert0 = f3{ t.v0[0], t.v0[1], t.v0[2] }, edge1 = f3{ t.e1[0], t.e1[1], t.e1[2] }, edge2 = f3{ t.e2[0], t.e2[1], t.e2[2] };
    const f3 tvec = f3_sub(f3{ ox, oy, oz }, vert0);
    const f3 qvec = f3_cross(tvec, edge1);
    TriCam c;
    c.e1[0] = edge1.x; c.e1[1] = edge1.y; c.e1[2] = edge1.z;
    c.e2[0] = edge2.x; c.e2[1] = edge2.y; c.e2[2] = edge2.z;
    c.tvec[0] = tvec.x; c.tvec[1] = tvec.y; c.tvec[2] = tvec.z;
    c.qvec[0] = qvec.x; c.qvec[1] = qvec.y; c.qvec[2] = qvec.z;
    c.e2_dot_qvec = f3_dot(edge2, qvec);
    c.pos = t.pos; c.pad[0] = c.pad[1] = 0u;
    a.tricam[i] = c;
}

// triangleIntersect, Triangle.h:66-107, against precomputed edges; keeps the lexicographic min of
// (t, position) -- the strict `<` of BVH.h:392 in DFS order.
__device__ __forceinline__ void leaf_test(const TraceArgs& a, uint32_t leaf, RayState& r)
{
    float4 v0 = a.v0[leaf], E1 = a.e1[leaf], E2 = a.e2[leaf];
    f3 vert0 = f3_from4(v0), edge1 = f3_from4(E1), edge2 = f3_from4(E2);
    f3 pvec = f3_cross(r.d, edge2);
    float det = f3_dot(edge1, pvec);
    bool flag = fabsf(det) >= FLT_EPSILON;
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(r.o, vert0);
    float u = f3_dot(tvec, pvec) * inv_det;
    flag = flag & (u < 1.0f);
    flag = flag & (u >= 0.0f);
    f3 qvec = f3_cross(tvec, edge1);
    float v = f3_dot(r.d, qvec) * inv_det;
    flag = flag & ((u + v) <= 1.0f);
    flag = flag & (v >= 0.0f);
    float t = f3_dot(edge2, qvec) * inv_det;      // "* flag" (Triangle.h:104) folded into the test below
    if (flag & (t > 0.0f)) {
        if (t < r.best_t) { r.best_t = t; r.best_leaf = leaf; }
        else if (t == r.best_t && r.best_leaf != NO_LEAF && a.pos[leaf] < a.pos[r.best_leaf]) r.best_leaf = leaf;
    }
}

// work item -> pixel.  Work items enumerate this rank's 64x64 tiles; inside a tile the 64 8x8
// blocks, inside a block the 64 pixels (one wave = one 8x8 block).
__device__ __forceinline__ bool work_to_pixel(const TraceArgs& a, uint32_t w, uint32_t& x, uint32_t& y, uint32_t& slot)
{
    uint32_t ordinal = w >> 12;                 // owned tile ordinal (4096 px per tile)
    uint32_t within = w & 4095u;
    uint32_t tile = rt_tile_of(a.rank, ordinal, a.nranks);
    uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    uint32_t blk = within >> 6, lp = within & 63u;
    uint32_t ly = ((blk >> 3) << 3) + (lp >> 3);
    uint32_t lx = ((blk & 7u) << 3) + (lp & 7u);
    x = tx * RT_TILE + lx; y = ty * RT_TILE + ly;
    slot = (ordinal << 12) + ly * RT_TILE + lx;
    return x < a.cam.resX && y < a.cam.resY;
}
// ... for a wave whose 64 lanes hold the 64 consecutive work items of ONE 8x8 block (every first-phase kernel): the tile is the same
// for all of them, and where it lies comes out of a table of this frame shape (one scalar load) instead of three integer divisions per
// lane (~80 vector instructions of the ~165 a wave spends outside its steps)
__device__ __forceinline__ bool work_to_pixel_block(const TraceArgs& a, uint32_t w, uint32_t& x, uint32_t& y, uint32_t& slot)
{
    if (a.tile_xy == nullptr) return work_to_pixel(a, w, x, y, slot);
    const uint32_t ordinal = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 12));
    const uint2 t0 = a.tile_xy[ordinal];
    const uint32_t within = w & 4095u;
    const uint32_t blk = within >> 6, lp = within & 63u;
    const uint32_t ly = ((blk >> 3) << 3) + (lp >> 3);
    const uint32_t lx = ((blk & 7u) << 3) + (lp & 7u);
    x = t0.x + lx; y = t0.y + ly;
    slot = (ordinal << 12) + ly * RT_TILE + lx;
    return x < a.cam.resX && y < a.cam.resY;
}

// loads the ray of work item w; false when the item is a slot outside the frame
template <int PRIMARY, bool BLOCK = true>            // BLOCK: the wave's lanes hold one 8x8 block's work items (work_to_pixel_block)
__device__ __forceinline__ bool load_ray(const TraceArgs& a, uint32_t w, RayState& r, bool& sx, bool& sy, bool& sz)
{
    if (PRIMARY) {
        uint32_t x, y, slot;
        if (!(BLOCK ? work_to_pixel_block(a, w, x, y, slot) : work_to_pixel(a, w, x, y, slot))) return false;
        generate_ray_xy(a.cam, x, y, r.o, r.d);
        r.pixel = y * a.cam.resX + x;
        r.out_index = a.compact ? slot : r.pixel;
    } else {
        // kernel generateContacts2, BVHKernels.cl:103-117
        const float4* p = reinterpret_cast<const float4*>(a.rays + w);
        float4 o = p[1], d = p[2];
        r.o = f3{ o.x, o.y, o.z };
        r.d = f3{ d.x, d.y, d.z };
        r.pixel = w; r.out_index = w;
    }
    r.work = w;
    r.inv = f3{ 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };   // INV_DIR_*, AABB.h:115-117
    sx = 0.0f > r.inv.x; sy = 0.0f > r.inv.y; sz = 0.0f > r.inv.z;
    r.best_t = FLT_MAX; r.best_leaf = NO_LEAF; r.best_pos = 0u;      // rank 0 = nothing ranks before it: no tie update without a hit
    return true;
}

// Calibration launches (the <.., 1> instances, never the frames that count) carry a deadline on the clock all XCDs share: a
// candidate strategy that has fallen clearly behind the best one so far stops walking -- its frame is thrown away anyway -- so that
// the first frame on a BVH costs a bounded multiple of a steady one whatever the hopeless candidates would have taken
// (round 2: up to 55 ms for one of them against 0.1-7.5 ms frames).  Compiled out of every other instance.
// A wave that stops here says so (TraceCtrl::expired, cleared per launch): the host must not take the elapsed time of a launch that
// was cut off for the time the strategy needs -- a deadline mapped too early (the host thread descheduled between its time stamp
// and the launch) would otherwise make a truncated candidate the fastest one.
template <int CAL>
__device__ __forceinline__ bool cal_expired(const TraceArgs& a)
{
    if (CAL != 1) return false;
    const bool expired = a.cal_deadline != 0ull && __builtin_amdgcn_s_memrealtime() > a.cal_deadline;       // (wave-uniform: scalar)
    if (expired && (threadIdx.x & 63u) == 0u) __hip_atomic_store((gu32*)&a.ctrl->expired, 1u, RLX_AGENT);
    return expired;
}
__global__ void k_device_clock(unsigned long long* out) { *out = __builtin_amdgcn_s_memrealtime(); }
void launch_device_clock(hipStream_t s, unsigned long long* d_out) { hipLaunchKernelGGL(k_device_clock, dim3(1), dim3(1), 0, s, d_out); }

// the camera-on-a-plane bits of this frame: computed by k_prepare_camera on the device, or handed over by a host that knows them
__device__ __forceinline__ uint32_t camera_plane_bits(const TraceArgs& a)
{
    if (a.search_root == 0u) return 0u;
    return (a.plane_bits_known & 0x80000000u) ? (a.plane_bits_known & 7u) : *a.tree_select;
}

// root record a ray starts from: the search tree's, unless there is none or the ray could produce 0 * inf in a slab test
// (camera coordinate on a box plane AND 1/direction not finite on that axis: zero, denormal or NaN direction component)
template <int PRIMARY>
__device__ __forceinline__ uint32_t ray_root(const TraceArgs& a, const RayState& r, uint32_t plane_bits)
{
    if (a.search_root == 0u) return 0u;
    if (!PRIMARY) {
        // arbitrary rays have arbitrary origins: no plane test is possible, so a ray may use the search tree only if none of
        // its slab products can be 0 * inf or underflow -- every 1/d finite and not smaller than ~1 in magnitude, which
        // normalised directions without a zero component satisfy
        const float ax = fabsf(r.inv.x), ay = fabsf(r.inv.y), az = fabsf(r.inv.z);
        // ... and a finite origin of moderate size: a NaN origin coordinate on y or z makes the reference's staged test forget its
        // earlier axes ("the first axis"), and (plane - origin) must not overflow
        // On the X axis an infinite 1/d is allowed: a 0 * inf there (origin on a box plane) is a NaN that the sign-selected tests of the
        // lane-per-ray and wave-per-ray kernels ignore (v_max / v_min), which is what the reference's staged test does with it (DESIGN.md
        // section 2, "the first axis"); every other product on that axis is +-inf with the right sign.  (A camera centred on its model hands
        // a whole pixel column of such rays over: the Sponza-class stand-in's 1 080 of them walked the 10 835-level reference tree, 0.5 ms.)
        const bool safe = (ax >= 0.99f) && (ay <= FLT_MAX) && (az <= FLT_MAX) && (ay >= 0.99f) && (az >= 0.99f) &&
                          (fabsf(r.o.x) <= 1e30f) && (fabsf(r.o.y) <= 1e30f) && (fabsf(r.o.z) <= 1e30f);
        return safe ? a.search_root : 0u;
    }
    const bool risky = ((plane_bits & 1u) && !(fabsf(r.inv.x) <= FLT_MAX)) || ((plane_bits & 2u) && !(fabsf(r.inv.y) <= FLT_MAX)) ||
                       ((plane_bits & 4u) && !(fabsf(r.inv.z) <= FLT_MAX));
    return risky ? 0u : a.search_root;
}

// Where the four-wide packet kernel sends a ray it cannot walk itself: RT_SENTINEL = nowhere (the ray walks with its packet), else the
// record the wave-per-ray kernel starts it at -- the reference root (ray_root), or, for a ray of a ray buffer whose 1/d is infinite on x,
// the search tree's root: that kernel's sign-selected test takes the NaN such a ray can produce, the packet's min / max form would not.
template <int PRIMARY>
__device__ __forceinline__ uint32_t packet_hand_over(const TraceArgs& a, const RayState& r, uint32_t plane_bits)
{
    const uint32_t root = ray_root<PRIMARY>(a, r, plane_bits);
    if (root == 0u) return 0u;
    if (!PRIMARY && !(fabsf(r.inv.x) <= FLT_MAX)) return root;
    return RT_SENTINEL;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Contact assembly, BVH.h:402-407 + BVHKernels.cl:95-96
__device__ __forceinline__ void store_contact(const TraceArgs& a, uint32_t out_index, uint32_t pixel, float t, uint32_t leaf)
{
    float4 nt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t material = 0;
    if (leaf != NO_LEAF) {
        float4 n = a.nrm[leaf];
        nt = make_float4(n.x, n.y, n.z, t);
        material = __float_as_uint(a.v0[leaf].w);
    }
    uint4 head = make_uint4(pixel, material, 0u, 0u);
    float4* o = reinterpret_cast<float4*>(a.out + out_index);
    store16_stream(o, *reinterpret_cast<float4*>(&head));
    store16_stream(o + 1, nt);
}

// ... and, in a fused frame, its colour: what k_shade computes from the contact just written, at the index k_shade writes it to
// (shade_kernels.hip: the frame's own pixel / compact slot; misses are zeros; the packed planes carry the hit in red's sign bit)
__device__ __forceinline__ void store_colour(const TraceArgs& a, uint32_t out_index, uint32_t pixel, float t, uint32_t leaf)
{
    float4 col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool hit = leaf != NO_LEAF && t > 0.0f;
    if (hit) {
        rt_contact c;
        const float4 n = a.nrm[leaf];
        c.pixelIndex = pixel; c.materialIndex = __float_as_uint(a.v0[leaf].w); c.pad[0] = 0; c.pad[1] = 0;
        c.normalAndDistance[0] = n.x; c.normalAndDistance[1] = n.y; c.normalAndDistance[2] = n.z; c.normalAndDistance[3] = t;
        col = shade_one(a.fused_scene, a.cam, c);
    }
    if (a.fused_plane_stride) {
        float* planes = reinterpret_cast<float*>(a.fused_rgba);
        if (hit) col.x = __uint_as_float(__float_as_uint(col.x) | 0x80000000u);
        store4_stream(planes + out_index, col.x);
        store4_stream(planes + (size_t)a.fused_plane_stride + out_index, col.y);
        store4_stream(planes + 2u * (size_t)a.fused_plane_stride + out_index, col.z);
    } else {
        store16_stream(a.fused_rgba + out_index, col);
    }
}

// ------------------------------------------------------------------------------------------------
// phase 0: one ray per lane
// ------------------------------------------------------------------------------------------------
template <int PRIMARY, int CAL>
__global__ __launch_bounds__(TRACE_BLOCK, 5) void k_trace(TraceArgs a)
{
    extern __shared__ uint32_t s_dynamic[];              // LDS_STACK x 64 words per wave of the workgroup (launch_tagged sizes it)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t (*stack)[64] = reinterpret_cast<uint32_t (*)[64]>(s_dynamic + wave * (LDS_STACK * 64u));
    const uint32_t wave_gid = blockIdx.x * (blockDim.x >> 6) + wave;     // (workgroups of one to TRACE_WAVES waves: launch_tagged)
    gu32* long_count = (gu32*)&a.ctrl->long_count;
    gu32* error_word = (gu32*)&a.ctrl->error;
    const uint32_t plane_bits = PRIMARY ? camera_plane_bits(a) & ~1u : 0u;       // (x: child_valid_signed)

    RayState r;
    r.o = f3{ 0, 0, 0 }; r.d = f3{ 0, 0, 1 }; r.inv = f3{ 0, 0, 1 };
    r.best_t = FLT_MAX; r.best_leaf = NO_LEAF; r.best_pos = 0u; r.work = 0; r.out_index = 0; r.pixel = 0;
    bool sx = false, sy = false, sz = false;
    uint32_t cur = RT_SENTINEL;       // node to visit: inner record index, RT_LEAF_FLAG|leaf, or sentinel
    uint32_t pending = NO_LEAF;       // parked leaf
    uint32_t sp = 0;                  // entries on this lane's stack
    uint32_t iters = 0;
    bool active = false;
    bool may_defer = true;            // false once the long-ray list turned out to be full

    // one wave per 64-ray block (an 8x8-pixel block of a primary frame): the hardware dispatcher is the work queue
    {
        const uint32_t w = (wave_gid << 6) + lane;
        if (w < a.total_work) {
            if (load_ray<PRIMARY>(a, w, r, sx, sy, sz)) {
                cur = ray_root<PRIMARY>(a, r, plane_bits);    // record 0 = the reference root (node n), or the search tree's root
                active = true;
            } else if (PRIMARY && a.compact) {
                // slot of a border tile that lies outside the frame: defined as "no contact"
                uint32_t x, y, slot;
                work_to_pixel_block(a, w, x, y, slot);
                float4* o = reinterpret_cast<float4*>(a.out + slot);
                o[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                o[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (a.fused_rgba != nullptr) store_colour(a, slot, 0u, 0.0f, NO_LEAF);
            }
        }
    }
    if (cal_expired<CAL>(a)) return;
    uint32_t cal_tick = 0;
    if (__ballot(active) != 0ull) {
        // ---------------- traversal steps until every ray of the block has finished or moved on ----------------
        for (;;) {
            if (CAL == 1 && (++cal_tick & 15u) == 0u && cal_expired<CAL>(a)) break;
            uint32_t extra = RT_SENTINEL;     // child that did not fit on the stack (forces a deferral)
            if (active) {
                iters++;
                if (cur != RT_SENTINEL && (cur & RT_LEAF_FLAG) == 0u && a.quad != nullptr && cur >= a.search_root && sp + 3u <= LDS_STACK) {
                    // inner node, four-wide: the 128-byte record holds the node's grandchildren, so one step descends two
                    // levels and the dependent chain (record -> tests -> next record) that this kernel waits on is half as long.
                    // Skipping the inner children's own boxes changes nothing only where the search-tree argument holds
                    // (DESIGN.md section 2), so this step is taken on search-tree records alone: rays that ray_root sent to the
                    // reference tree (a slab product could be 0 * inf) walk it record by record, box by box.
                    // Up to three entries are pushed (only while they fit; otherwise the binary step below).
                    const float4* p = reinterpret_cast<const float4*>(a.quad + cur);
                    const float4 g0 = p[0], g1 = p[1], g2 = p[2], g3 = p[3], g4 = p[4], g5 = p[5], g6 = p[6], g7 = p[7];
                    const uint32_t i0 = __float_as_uint(g0.w), i1 = __float_as_uint(g1.w), i2 = __float_as_uint(g2.w), i3 = __float_as_uint(g3.w);
                    // (rays on search-tree records can produce a NaN slab product on x only, ray_root: the sign-selected pure-VALU form
                    //  of the test; primary rays read the camera-relative copy of the records)
                    auto slot_valid = [&](const float4& lo, const float4& hi, float threshold) {
                        if (PRIMARY) return child_valid_delta_signed(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, r, sx, sy, sz, threshold);
                        return child_valid_general_signed(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, r, sx, sy, sz);
                    };
                    uint32_t c0 = slot_valid(g0, g1, g4.w) ? i0 : RT_SENTINEL;
                    uint32_t c1 = slot_valid(g2, g3, g5.w) ? i1 : RT_SENTINEL;
                    uint32_t c2 = slot_valid(g4, g5, g6.w) ? i2 : RT_SENTINEL;
                    uint32_t c3 = slot_valid(g6, g7, g7.w) ? i3 : RT_SENTINEL;
                    if (pending == NO_LEAF) {                      // park one reached leaf right away (no stack round trip)
                        if (c0 != RT_SENTINEL && (c0 & RT_LEAF_FLAG)) { pending = c0 & ~RT_LEAF_FLAG; c0 = RT_SENTINEL; }
                        else if (c1 != RT_SENTINEL && (c1 & RT_LEAF_FLAG)) { pending = c1 & ~RT_LEAF_FLAG; c1 = RT_SENTINEL; }
                        else if (c2 != RT_SENTINEL && (c2 & RT_LEAF_FLAG)) { pending = c2 & ~RT_LEAF_FLAG; c2 = RT_SENTINEL; }
                        else if (c3 != RT_SENTINEL && (c3 & RT_LEAF_FLAG)) { pending = c3 & ~RT_LEAF_FLAG; c3 = RT_SENTINEL; }
                    }
                    uint32_t nxt = RT_SENTINEL;
#define RT_TAKE(c) if (c != RT_SENTINEL) { if (nxt == RT_SENTINEL) nxt = c; else { stack[sp][lane] = c; sp++; } }
                    RT_TAKE(c0) RT_TAKE(c1) RT_TAKE(c2) RT_TAKE(c3)
#undef RT_TAKE
                    if (nxt != RT_SENTINEL) cur = nxt;
                    else if (sp == 0) cur = RT_SENTINEL;
                    else { sp--; cur = stack[sp][lane]; }
                } else if (cur != RT_SENTINEL && (cur & RT_LEAF_FLAG) == 0u) {
                    // inner node: one 64-byte record
                    const float4* p = reinterpret_cast<const float4*>(a.trav + cur);
                    float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                    uint32_t ca = __float_as_uint(q0.w), cb = __float_as_uint(q1.w);
                    const uint32_t marks = __float_as_uint(q2.w);
                    const bool on_search = a.search_root != 0u && cur >= a.search_root;
                    bool aValid = child_valid_on<PRIMARY>(on_search, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r, sx, sy, sz, (marks & 1u) != 0u);
                    bool bValid = child_valid_on<PRIMARY>(on_search, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r, sx, sy, sz, (marks & 2u) != 0u);
                    if (aValid & bValid) {
                        const bool a_leaf = (ca & RT_LEAF_FLAG) != 0u, b_leaf = (cb & RT_LEAF_FLAG) != 0u;
                        if (pending == NO_LEAF && (a_leaf | b_leaf)) {
                            // one child is a leaf and the parking slot is free: park it right away and walk on with the
                            // other child -- no stack round trip (the common step along a duplicate-code chain)
                            pending = (a_leaf ? ca : cb) & ~RT_LEAF_FLAG;
                            cur = a_leaf ? cb : ca;
                        } else {
                            cur = ca;
                            if (sp < LDS_STACK) { stack[sp][lane] = cb; sp++; } else extra = cb;
                        }
                    } else if (aValid | bValid) cur = aValid ? ca : cb;
                    else if (sp == 0) cur = RT_SENTINEL;
                    else { sp--; cur = stack[sp][lane]; }
                }
                // park a reached leaf if the slot is free, and move on
                if (extra == RT_SENTINEL && cur != RT_SENTINEL && (cur & RT_LEAF_FLAG) != 0u && pending == NO_LEAF) {
                    pending = cur & ~RT_LEAF_FLAG;
                    if (sp == 0) cur = RT_SENTINEL;
                    else { sp--; cur = stack[sp][lane]; }
                }
            }
            // a lane that needs its slot again (second leaf) or is about to finish forces the
            // wave through the triangle test for every parked leaf
            bool must = active && pending != NO_LEAF && (cur == RT_SENTINEL || (cur & RT_LEAF_FLAG) != 0u);
            if (__ballot(must) != 0ull) {
                if (active && pending != NO_LEAF) { leaf_test(a, pending, r); pending = NO_LEAF; }
            }

            // ---- defer long rays to phase 1 (budget exceeded, or stack full) ----
            bool want_defer = active && cur != RT_SENTINEL &&
                              (extra != RT_SENTINEL || (may_defer && iters >= a.defer_budget));
            uint64_t deferring = __ballot(want_defer);
            if (deferring != 0ull) {
                uint32_t base = 0;
                int leader = __ffsll((unsigned long long)deferring) - 1;
                if ((int)lane == leader) base = __hip_atomic_fetch_add(long_count, (uint32_t)__popcll(deferring), RLX_AGENT);
                base = (uint32_t)__shfl((int)base, leader, 64);
                if (want_defer) {
                    uint32_t idx = base + lane_rank(deferring);
                    if (idx < a.long_cap) {
                        uint32_t* rec = a.long_list + (uint64_t)idx * LONG_REC_WORDS;
                        rec[LR_WORK] = r.work; rec[LR_CUR] = cur; rec[LR_SP] = sp; rec[LR_BEST_LEAF] = r.best_leaf;
                        rec[LR_BEST_T] = __float_as_uint(r.best_t); rec[LR_EXTRA] = extra; rec[LR_PENDING] = pending; rec[LR_PARTIAL] = 0u;
                        for (uint32_t k = 0; k < sp; k++) rec[LR_STACK + k] = stack[k][lane];
                        active = false;
                    } else if (extra != RT_SENTINEL) {
                        TRACE_FAIL(error_word, 1);     // nowhere to put the sub-tree: frame invalid
                        active = false;
                    } else {
                        may_defer = false;                                 // list full: finish the ray here
                    }
                }
            }

            // ---- retire finished rays ----
            if (active && cur == RT_SENTINEL && pending == NO_LEAF) {
                store_contact(a, r.out_index, r.pixel, r.best_t, r.best_leaf);
                if (PRIMARY && a.fused_rgba != nullptr) store_colour(a, r.out_index, r.pixel, r.best_t, r.best_leaf);   // fused frame of the lane-per-ray strategy: launch_tagged
                active = false;
            }
            if (__ballot(active) == 0ull) break;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// packet traversal: one 8x8-pixel block of primary rays per wave, ONE node at a time for all 64 rays
// ------------------------------------------------------------------------------------------------
// The lane-per-ray kernel is bound by the vector memory pipe: every lane gathers its own 64-byte record
// (measured ~190 address-processing cycles per wave step and CU, insensitive to occupancy; HBM idle).
// Neighbouring primary rays reach almost the same nodes, only at different times.  Here the wave walks the
// tree ONCE for its 64 rays: the node is wave-uniform, so its record comes through the SCALAR cache into
// SGPRs (no vector-memory traffic at all) and each lane only evaluates its own ray against it.  A lane takes
// part in a node iff its own ray found every ancestor valid (per-lane `in` flag; the DFS stack holds
// (node, lane mask) pairs), so every ray still sees exactly the leaves the reference's walk reaches and keeps
// the lexicographic (t, position) minimum: the contacts are bit-identical.
typedef const __attribute__((address_space(4))) float cf4;        // constant address space: uniform loads become s_load
typedef const __attribute__((address_space(4))) uint32_t cu32;
__device__ __forceinline__ cf4* as_const(const void* p) { return (cf4*)(unsigned long long)p; }
__device__ __forceinline__ float4 ld_const4(const void* p) { cf4* f = as_const(p); return make_float4(f[0], f[1], f[2], f[3]); }
struct NodeRec { float4 q0, q1, q2, q3; };
// one 64-byte record through the scalar path (the index is wave-uniform)
__device__ __forceinline__ NodeRec ld_record(const TravNode* trav, uint32_t node)
{
    node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);       // uniform by construction: keeps the record in SGPRs
    const float4* p = reinterpret_cast<const float4*>(trav + node);
    NodeRec r;
    r.q0 = ld_const4(p); r.q1 = ld_const4(p + 1); r.q2 = ld_const4(p + 2); r.q3 = ld_const4(p + 3);
    return r;
}
#ifdef RTMI_TEST_PACKET_STACK
#define PACKET_STACK RTMI_TEST_PACKET_STACK      // librtmi_teststack.so: every deep block overflows into the global areas
#else
#define PACKET_STACK 96u
#endif
#ifdef RTMI_TEST_PACKET_STACK
#define PACKET_VSTACK 2u           // test twin: the lane-resident part of the four-wide kernel's stack overflows at once as well
#else
#define PACKET_VSTACK 64u          // k_trace_packet4: stack entries held in lanes of vector registers
#endif
// Three scalar values -> lane `l` of three vector registers (v_writelane_b32; the builtin is not exposed by this compiler).  The
// lane select goes through M0 because a VOP3 instruction may read only one SGPR over the constant bus.  Synchronous: nothing is
// in flight behind it; the s_nop covers the wait state between the write of M0 and its use as lane select.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"          // M0 is a reserved register: the clobber is a note to the reader
__device__ __forceinline__ void lane_write3(uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t a, uint32_t b, uint32_t c, uint32_t l)
{
    // (uniform by construction; said again for the compiler, which cannot always tell and must put them in scalar registers)
    a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a); b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c); l = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
    asm("s_mov_b32 m0, %6\n\ts_nop 1\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
        : "+v"(v0), "+v"(v1), "+v"(v2) : "s"(a), "s"(b), "s"(c), "s"(l) : "m0");
}
#pragma clang diagnostic pop
#define PACKET_LOW_LANES 4u        // "a handful of lanes"
#define PACKET_LOW_RUN 24u         // consecutive low-occupancy steps before the block sheds sub-trees

__device__ __forceinline__ void leaf_test_uniform(const TraceArgs& a, uint32_t leaf, RayState& r)
{
    float4 v0 = ld_const4(a.v0 + leaf), E1 = ld_const4(a.e1 + leaf), E2 = ld_const4(a.e2 + leaf);
    f3 vert0 = f3_from4(v0), edge1 = f3_from4(E1), edge2 = f3_from4(E2);
    f3 pvec = f3_cross(r.d, edge2);
    float det = f3_dot(edge1, pvec);
    bool flag = fabsf(det) >= FLT_EPSILON;
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(r.o, vert0);
    float u = f3_dot(tvec, pvec) * inv_det;
    flag = flag & (u < 1.0f);
    flag = flag & (u >= 0.0f);
    f3 qvec = f3_cross(tvec, edge1);
    float v = f3_dot(r.d, qvec) * inv_det;
    flag = flag & ((u + v) <= 1.0f);
    flag = flag & (v >= 0.0f);
    float t = f3_dot(edge2, qvec) * inv_det;
    if (flag & (t > 0.0f)) {
        if (t < r.best_t) { r.best_t = t; r.best_leaf = leaf; }
        else if (t == r.best_t && r.best_leaf != NO_LEAF && a.pos[leaf] < a.pos[r.best_leaf]) r.best_leaf = leaf;
    }
}

// the same for the lanes of `reached` only, without divergent control flow: every lane computes, the update is a select; the
// tie on t (the rule of the depth-first rank) is rare and sits behind a wave-uniform test
__device__ __forceinline__ void leaf_test_record(const TriRec* __restrict__ trirec, uint32_t leaf, RayState& r, bool reached)
{
    // the packed record through the scalar path: one address, twelve dwords
    const float4* p = reinterpret_cast<const float4*>(trirec + leaf);
    const float4 w0 = ld_const4(p), w1 = ld_const4(p + 1), w2 = ld_const4(p + 2);
    const f3 vert0 = f3{ w0.x, w0.y, w0.z }, edge1 = f3{ w0.w, w1.x, w1.y }, edge2 = f3{ w1.z, w1.w, w2.x };
    const uint32_t pos = __float_as_uint(w2.y);
    f3 pvec = f3_cross(r.d, edge2);
    float det = f3_dot(edge1, pvec);
    bool flag = reached & (fabsf(det) >= FLT_EPSILON);
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(r.o, vert0);
    float u = f3_dot(tvec, pvec) * inv_det;
    flag = flag & (u < 1.0f);
    flag = flag & (u >= 0.0f);
    f3 qvec = f3_cross(tvec, edge1);
    float v = f3_dot(r.d, qvec) * inv_det;
    flag = flag & ((u + v) <= 1.0f);
    flag = flag & (v >= 0.0f);
    float t = f3_dot(edge2, qvec) * inv_det;
    flag = flag & (t > 0.0f);
    // lexicographic minimum of (t, depth-first rank): the strict `<` of BVH.h:392 in the reference's walk order
    const bool better = flag & ((t < r.best_t) | ((t == r.best_t) & (pos < r.best_pos)));
    r.best_t = better ? t : r.best_t;
    r.best_leaf = better ? leaf : r.best_leaf;
    r.best_pos = better ? pos : r.best_pos;
}

// ... of a primary ray, against the camera-relative record: the ray contributes its direction only
__device__ __forceinline__ void leaf_test_camera(const TriCam* __restrict__ tricam, uint32_t leaf, RayState& r, bool reached)
{
#ifdef RTMI_EXP_NO_LEAF
    return;                      // EXPERIMENT (wrong contacts): what the walk costs without its triangle tests
#endif
    const float4* p = reinterpret_cast<const float4*>(tricam + leaf);
    const float4 w0 = ld_const4(p), w1 = ld_const4(p + 1), w2 = ld_const4(p + 2), w3 = ld_const4(p + 3);
#ifdef RTMI_EXP_LEAF_LOAD_ONLY
    asm volatile("" :: "s"(w0.x), "s"(w1.x), "s"(w2.x), "s"(w3.x));      // EXPERIMENT (wrong contacts): the record arrives, nothing is computed
    return;
#endif
    const f3 edge1 = f3{ w0.x, w0.y, w0.z }, edge2 = f3{ w0.w, w1.x, w1.y }, tvec = f3{ w1.z, w1.w, w2.x }, qvec = f3{ w2.y, w2.z, w2.w };
    const uint32_t pos = __float_as_uint(w3.y);
    f3 pvec = f3_cross(r.d, edge2);
    float det = f3_dot(edge1, pvec);
    bool flag = reached & (fabsf(det) >= FLT_EPSILON);
    float inv_det = 1.0f / det;
    float u = f3_dot(tvec, pvec) * inv_det;
    flag = flag & (u < 1.0f);
    flag = flag & (u >= 0.0f);
    float v = f3_dot(r.d, qvec) * inv_det;
    flag = flag & ((u + v) <= 1.0f);
    flag = flag & (v >= 0.0f);
    float t = w3.x * inv_det;
    flag = flag & (t > 0.0f);
    const bool better = flag & ((t < r.best_t) | ((t == r.best_t) & (pos < r.best_pos)));
    r.best_t = better ? t : r.best_t;
    r.best_leaf = better ? leaf : r.best_leaf;
    r.best_pos = better ? pos : r.best_pos;
}

// Overflow areas of the packet stacks.  One wave per block means far more waves than areas, so a wave takes an area only
// when its LDS stack overflows (rare) and gives it back when it ends: a bitmap with one bit per area, as many areas as the
// device can hold resident waves, so a free bit always exists; the search is bounded all the same and reports error 5.
// Wave-uniform; returns the area or nullptr.
__device__ __forceinline__ uint32_t* spill_acquire(const TraceArgs& a, uint32_t wave_gid, uint32_t lane, uint32_t& slot)
{
    uint32_t got = 0xFFFFFFFFu;
    if (lane == 0) {
        const uint32_t words = a.packet_spill_slots >> 5;
        uint32_t w = (wave_gid * 2654435761u >> 8) % words;
        for (uint32_t tries = 0; tries < 64u * words && got == 0xFFFFFFFFu; tries++) {
            const uint32_t used = __hip_atomic_load((gu32*)a.packet_spill + w, RLX_AGENT);
            if (used != 0xFFFFFFFFu) {
                const uint32_t bit = (uint32_t)__builtin_ctz(~used);
                const uint32_t old = __hip_atomic_fetch_or((gu32*)a.packet_spill + w, 1u << bit, RLX_AGENT);
                if ((old & (1u << bit)) == 0u) { got = (w << 5) + bit; __hip_atomic_fetch_add((gu32*)&a.ctrl->spill_taken, 1u, RLX_AGENT); }
            } else {
                w = w + 1u == words ? 0u : w + 1u;
            }
        }
        if (got == 0xFFFFFFFFu) TRACE_FAIL((gu32*)&a.ctrl->error, 5);
    }
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
    if (slot == 0xFFFFFFFFu) return nullptr;
    return a.packet_spill + (a.packet_spill_slots >> 5) + (uint64_t)slot * a.packet_spill_entries * 3u;
}
__device__ __forceinline__ void spill_release(const TraceArgs& a, uint32_t lane, uint32_t slot)
{
    if (slot != 0xFFFFFFFFu && lane == 0) __hip_atomic_fetch_and((gu32*)a.packet_spill + (slot >> 5), ~(1u << (slot & 31u)), RLX_AGENT);
}

template <int PRIMARY, int CAL>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_packet(TraceArgs a)
{
    __shared__ uint32_t s_node[TRACE_WAVES][PACKET_STACK];
    __shared__ uint32_t s_mlo[TRACE_WAVES][PACKET_STACK];
    __shared__ uint32_t s_mhi[TRACE_WAVES][PACKET_STACK];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wave_gid = blockIdx.x * (blockDim.x >> 6) + wave;
    uint32_t* st_node = s_node[wave]; uint32_t* st_lo = s_mlo[wave]; uint32_t* st_hi = s_mhi[wave];
    // deep stacks spill to a global area (3 words per entry), taken on the first overflow
    uint32_t* spill = nullptr;
    uint32_t spill_slot = 0xFFFFFFFFu;

    const uint32_t plane_bits = PRIMARY ? camera_plane_bits(a) : 0u;
    // one wave per 8x8-pixel block: the hardware dispatcher is the work queue
    if ((wave_gid << 6) < a.total_work) {
        const uint32_t w = (wave_gid << 6) + lane;
        RayState r;
        bool sx = false, sy = false, sz = false;
        r.o = f3{ 0, 0, 0 }; r.d = f3{ 0, 0, 1 }; r.inv = f3{ 0, 0, 1 }; r.best_t = FLT_MAX; r.best_leaf = NO_LEAF; r.best_pos = 0u;
        r.work = w; r.out_index = 0; r.pixel = 0;
        bool ok = w < a.total_work && load_ray<PRIMARY>(a, w, r, sx, sy, sz);
        if (PRIMARY && a.compact && !ok && w < a.total_work) {
            uint32_t x, y, slot;
            work_to_pixel(a, w, x, y, slot);
            float4* o = reinterpret_cast<float4*>(a.out + slot);
            o[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            o[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        bool in = ok;
        bool split = false;                              // some sub-tree of this lane's ray went to phase 1
        uint32_t sp = 0;
        uint32_t node = 0;                               // wave-uniform: root record
        if (PRIMARY && a.search_root != 0u) {
            // the block walks the search tree; lanes whose ray must use the reference tree (ray_root) wait on the stack
            // as an ordinary (node, lane mask) entry with the reference root
            const bool on_ref = ok && ray_root<PRIMARY>(a, r, plane_bits) == 0u;
            const uint64_t m_ref = __ballot(on_ref), m_search = __ballot(ok && !on_ref);
            if (m_search != 0ull) {
                node = a.search_root; in = ok && !on_ref;
                if (m_ref != 0ull) {
                    if (lane == 0) { st_node[0] = 0u; st_lo[0] = (uint32_t)m_ref; st_hi[0] = (uint32_t)(m_ref >> 32); }
                    sp = 1;
                }
            }
        }
        uint32_t low_run = 0;                            // consecutive steps with very few lanes taking part
        uint32_t steps = 0;                              // nodes this block has walked
        bool walking = __ballot(in) != 0ull;
        NodeRec cur = ld_record(a.trav, node);           // record of `node`, kept in registers across steps
        if (cal_expired<CAL>(a)) walking = false;
        while (walking) {
            if (CAL == 1 && (steps & 15u) == 15u && cal_expired<CAL>(a)) break;
            // A block that keeps walking with a handful of lanes wastes the wave, and a block that walks thousands of
            // nodes sets the frame time.  After PACKET_LOW_RUN low-occupancy steps the current node -- after `defer_budget`
            // steps everything the block has left (current node and every stack entry) -- is handed, per taking-part lane,
            // to the wave-per-ray kernel as a (ray, sub-tree) task: long duplicate-code chains are walked 64 records at a
            // time there.  Parts of such a split ray meet in a 64-bit atomicMin on (t, position); k_resolve_split writes
            // the contact.
            {
                uint64_t mi = __ballot(in);
                uint32_t n_in = (uint32_t)__popcll(mi);
                low_run = n_in <= PACKET_LOW_LANES ? low_run + 1u : 0u;
                steps++;
                if (((low_run >= PACKET_LOW_RUN && steps * 4u > a.defer_budget) || steps > a.defer_budget) && a.long_cap != 0u &&
                    a.defer_budget < (1u << 20)) {                   // "no budget": nothing is ever handed over (the launcher skips phase 1)
                    uint32_t base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add((gu32*)&a.ctrl->long_count, n_in, RLX_AGENT);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    bool room = base + n_in <= a.long_cap;            // wave-uniform: all or nothing
                    if (room) {
                        if (in) {
                            if (!split) {
                                split = true;
                                uint32_t si = __hip_atomic_fetch_add((gu32*)&a.ctrl->split_count, 1u, RLX_AGENT);
                                if (si < a.long_cap) a.split_list[si] = w;
                                __hip_atomic_exchange((gu64*)a.keys + w, KEY_NONE, RLX_AGENT);
                            }
                            uint32_t* rec = a.long_list + (uint64_t)(base + lane_rank(mi)) * LONG_REC_WORDS;
                            rec[LR_WORK] = w; rec[LR_CUR] = node; rec[LR_SP] = 0u; rec[LR_BEST_LEAF] = NO_LEAF;
                            rec[LR_BEST_T] = __float_as_uint(FLT_MAX); rec[LR_EXTRA] = RT_SENTINEL; rec[LR_PENDING] = NO_LEAF;
                            rec[LR_PARTIAL] = 1u;
                        }
                        // continue with the next stack entry
                        if (sp == 0u) { walking = false; continue; }
                        sp--;
                        uint32_t nn, lo, hi;
                        if (sp < PACKET_STACK) { nn = st_node[sp]; lo = st_lo[sp]; hi = st_hi[sp]; }
                        else if (spill) { const uint32_t* e = spill + (uint64_t)(sp - PACKET_STACK) * 3u; nn = e[0]; lo = e[1]; hi = e[2]; }
                        else { nn = 0u; lo = 0u; hi = 0u; }                  // the push failed (error word set): an entry nobody wants
                        node = (uint32_t)__builtin_amdgcn_readfirstlane((int)nn);
                        in = ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u) != 0u;
                        cur = ld_record(a.trav, node);
                        continue;
                    }
                    // list full: keep walking here.  Slots this reservation still owns below the capacity must not
                    // be left with stale bytes (phase 1 reads every record below min(count, capacity)): mark them void
                    if (in) {
                        uint32_t idx = base + lane_rank(mi);
                        if (idx < a.long_cap) a.long_list[(uint64_t)idx * LONG_REC_WORDS + LR_WORK] = 0xFFFFFFFFu;
                    }
                    low_run = 0; steps = 0;
                }
            }
            // the record of `node` is in registers (cur); both children's records are requested NOW, before the box
            // tests, so that the scalar-cache / L2 latency of the next step runs under this step's arithmetic
            // (a leaf child requests record 0 instead: harmless, never used)
            const float4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2, q3 = cur.q3;
            const uint32_t ca = __float_as_uint(q0.w), cb = __float_as_uint(q1.w);
            const NodeRec recA = ld_record(a.trav, (ca & RT_LEAF_FLAG) ? 0u : ca);
            const NodeRec recB = ld_record(a.trav, (cb & RT_LEAF_FLAG) ? 0u : cb);
            const uint32_t marks = __float_as_uint(q2.w);
            bool aV = false, bV = false;
            if (in) {
                const bool on_search = a.search_root != 0u && node >= a.search_root;          // wave-uniform
                aV = child_valid_on<PRIMARY>(on_search, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r, sx, sy, sz, (marks & 1u) != 0u);
                bV = child_valid_on<PRIMARY>(on_search, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r, sx, sy, sz, (marks & 2u) != 0u);
            }
            uint64_t mA = __ballot(aV), mB = __ballot(bV);
            if ((ca & RT_LEAF_FLAG) != 0u) { if (mA != 0ull) { if (aV) leaf_test_uniform(a, ca & ~RT_LEAF_FLAG, r); } mA = 0ull; }
            if ((cb & RT_LEAF_FLAG) != 0u) { if (mB != 0ull) { if (bV) leaf_test_uniform(a, cb & ~RT_LEAF_FLAG, r); } mB = 0ull; }
            if (mA != 0ull && mB != 0ull) {
                // both inner children wanted: B waits on the stack with the lanes that want it
                if (sp < PACKET_STACK) {
                    if (lane == 0) { st_node[sp] = cb; st_lo[sp] = (uint32_t)mB; st_hi[sp] = (uint32_t)(mB >> 32); }
                } else if (a.packet_spill && sp - PACKET_STACK < a.packet_spill_entries && (spill || (spill = spill_acquire(a, wave_gid, lane, spill_slot)))) {
                    if (lane == 0) { uint32_t* e = spill + (uint64_t)(sp - PACKET_STACK) * 3u; e[0] = cb; e[1] = (uint32_t)mB; e[2] = (uint32_t)(mB >> 32); }
                } else {
                    if (lane == 0) TRACE_FAIL((gu32*)&a.ctrl->error, 3);     // frame invalid
                }
                sp++;
                node = ca; in = aV; cur = recA;
            } else if (mA != 0ull) { node = ca; in = aV; cur = recA; }
            else if (mB != 0ull) { node = cb; in = bV; cur = recB; }
            else {
                if (sp == 0u) { walking = false; }
                else {
                    sp--;
                    uint32_t nn, lo, hi;
                    if (sp < PACKET_STACK) { nn = st_node[sp]; lo = st_lo[sp]; hi = st_hi[sp]; }
                    else if (spill) { const uint32_t* e = spill + (uint64_t)(sp - PACKET_STACK) * 3u; nn = e[0]; lo = e[1]; hi = e[2]; }
                        else { nn = 0u; lo = 0u; hi = 0u; }                  // the push failed (error word set): an entry nobody wants
                    node = (uint32_t)__builtin_amdgcn_readfirstlane((int)nn);
                    in = ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u) != 0u;
                    cur = ld_record(a.trav, node);
                }
            }
        }
        if (ok) {
            store_contact(a, r.out_index, r.pixel, r.best_t, r.best_leaf);
            if (split && r.best_leaf != NO_LEAF)
                __hip_atomic_fetch_min((gu64*)a.keys + w, ((unsigned long long)__float_as_uint(r.best_t) << 32) | a.pos[r.best_leaf], RLX_AGENT);
        }

    }
    spill_release(a, lane, spill_slot);
}

// ------------------------------------------------------------------------------------------------
// packet kernel on the four-wide records: one wave step tests up to four (grand)children and descends two levels of the
// binary hierarchy, which halves the dependent chain  record -> box tests -> ballots -> next record  that bounds the
// binary packet kernel.  Nothing is handed over to the wave-per-ray kernel here (the calibration chooses this kernel
// only where whole blocks are cheap); everything else is as in k_trace_packet.
// ------------------------------------------------------------------------------------------------
// Where a wave step of this kernel spends its cycles (variant 2 below stamps four phases with s_memtime; ManFigure 1080p /
// atrium 4K, cycles per step): waiting for the record 543 / 210, four box tests + ballots 1 329 / 1 385, leaf tests 532 / 480,
// stack + choosing the next node 767 / 959 -- about 3 000 cycles per step at ~3 (ManFigure, tail of long blocks) and ~7
// (atrium) waves per SIMD alike (profiles/r02_packet4_phase_cycles.json).  Tried and NOT kept (round 2): the record as two
// s_load_dwordx16 in one round trip, the triangle data of leaf slots requested before the box tests, and the first inner
// slot's record requested speculatively.  Through inline assembly the requests are invisible to the compiler, which spilled
// the destination SGPRs of loads still in flight (a memory fault on the first frame); as ordinary loads the second record
// does not fit the SGPR file next to the first (106 SGPRs, 340 lane spills), and even without the speculation the wide loads
// cost 64 lane spills and 41 more VGPRs: ManFigure 0.667 -> 1.109 ms, atrium 18.3 -> 32.2 ms.  The compiler's lazy dword loads
// hit the scalar cache after the first one and are the better schedule.  Also tried after the box tests had become pure VALU
// (the record wait was then the longest phase, ~900 of ~2 900 cycles on ManFigure): all four inner slots' records requested
// before the tests with ONE vector load (lane 8 k + j takes the j-th 16 bytes of slot k's record, four VGPRs) and the chosen
// record read out of those lanes with 32 v_readlane after the tests -- no gain either (ManFigure 0.56-0.59 -> 0.59 ms, atrium
// 15.3 -> 15.6 ms): 106 SGPRs again (the pop path then has to materialise whole records), and the wait does not go away.
#ifndef RTMI_P4_VARIANT
#define RTMI_P4_VARIANT 0
#endif
// -DRTMI_P4_VARIANT=2: cycle stamps (s_memtime) around the four phases of a wave step, summed per launch in units of 64 cycles
// into ctrl->long_steps (record wait), long_steps_max (box tests + ballots), long_nodes (leaf tests), spill_taken (stack / next
// node), split_count (steps); read with tools/p4_time.py, the stamps and the atomics slow the kernel down (x 3).
// -DRTMI_P4_VARIANT=3: (start, end, steps) per wave on the 100 MHz clock all XCDs share, written to the key buffer and from
// there to the file RTMI_P4_WAVES names (tools/p4_timeline.py): no atomics, the launch takes what it takes without them.
#if RTMI_P4_VARIANT == 2
#define P4_STAMP(T) const unsigned long long T = __builtin_amdgcn_s_memtime()
#define P4_KEEP(...) asm volatile("" :: __VA_ARGS__)
#define P4_PIN(x) asm volatile("" : "+v"(x))          // what follows and uses x stays behind the stamp before it
#else
#define P4_STAMP(T)
#define P4_KEEP(...)
#define P4_PIN(x)
#endif
// SHARE = 1: a wave that has walked `defer_budget` steps and still has sub-trees waiting publishes its whole stack as TASKS
// (block, node, lane mask) that any wave may take: waves that have finished their own block look at the task queues before
// they exit, and helper waves at the end of the grid -- no block of their own -- keep taking tasks until nothing is running any
// more.  A frame's time was the time of its slowest wave (ManFigure: 633 steps against a mean of 51, Tenryuu 1 979 against 23,
// atrium 12 544 against 473; -DRTMI_P4_VARIANT=3), not the work divided by the machine.
// The reference walks every sub-tree a ray's boxes admit, whatever was hit before (BVH.h:343-389), so sub-trees are
// independent: the parts of a shared block meet in the 64-bit atomic minimum on (t bits, depth-first position) that the split
// rays of the other kernels use, and the helpers write those blocks' contacts once everything has finished.
//   * progress: block waves never wait (they take a task only if one is there); helpers wait, but they are far fewer than the
//     machine's wave slots, so every block wave is dispatched whatever the helpers do;
//   * exit: the first helper wave only watches: finished block waves == all of them and tasks finished == tasks published,
//     the three sums over the shards read in that order, each complete before the next starts.  The counters only grow, so
//     equality means that at the end of the second read nothing was running, and only running waves publish.  It then raises
//     the `finished` word of every shard, which the other helpers read together with the queue word they are polling.  Every
//     wave reaches its exit;
//   * visibility: everything another wave reads during the launch (entries, keys, split list, counters) is written with
//     device-scope atomic stores / read with device-scope atomic loads, and a wave waits for its stores (vmcnt) before it
//     writes an entry's valid mark or bumps a finishing counter.  No cache-wide invalidate or write-back is involved: fences
//     at device scope cost every wave on the XCD its L2 contents (measured: a frame 2 000 times slower);
//   * registers: what the steps of the walk need of the kernel arguments (records, triangles, budget) stays in scalar
//     registers; everything else is read again from the kernel-argument segment where it is used (rare_args), otherwise the
//     arguments of the rare paths stay live across the walk and the kernel spills.
// One 16-byte access at device scope (the sc1 bit the compiler puts on its device-scope atomic loads and stores; it has no
// 16-byte atomic type).  An aligned 16-byte access is one request to one cache line, so an entry is seen whole or not at all.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_device(uint32_t* p, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    u32x4 v = { x, y, z, w };
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 load16_device(const uint32_t* p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// "My stores and atomics have been performed": every vector-memory operation of this wave issued so far is acknowledged.  What
// another wave reads during the launch is written with sc1 stores / device-scope atomics (performed at the device-coherent level)
// and read with sc1 loads, so this wait IS the release, and no acquire-side invalidate is needed: nothing shared is read through
// a cached (non-sc1) load.  Written as the instruction itself: a workgroup-scope release fence compiles to `s_waitcnt lgkmcnt(0)`
// only on gfx950 (the vmcnt wait used to come from a neighbouring returning atomic, by luck of the schedule), and an agent-scope
// fence writes back / invalidates caches for every wave of the CU (measured: a frame 2 000 times slower).
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
// the kernel's (only) argument, read from the kernel-argument segment at the point of use: the pointer is made opaque so that
// the loads can neither be merged with earlier ones nor be moved out of the rarely executed block they stand in
// (SITE makes the statements differ: identical ones in different branches get merged behind a select of their inputs, which is
// no longer a scalar value as far as the compiler can tell)
template <int SITE>
__device__ __forceinline__ const TraceArgs& rare_args()
{
    const TraceArgs* p = (const TraceArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; kernel arguments, site %1" : "+s"(p) : "n"(SITE));
    return *p;
}

template <int PRIMARY, int CAL, int SHARE, int OCT, int FUSE>
#ifndef P4_MIN_WAVES
#define P4_MIN_WAVES 8
#endif
// (the instances without sharing: same bound.  "6" / "7" were measured when the fused instance still spilled 48 scalar registers per
//  wave: 106 / 94 scalar registers, bench +1.5 %, the 4K frame 0.38 -> 0.42 / 0.395 ms; the spills are gone since the fused frame's
//  shading arguments are read where they are used)
#ifndef P4_MIN_WAVES_WHOLE
#define P4_MIN_WAVES_WHOLE 8
#endif
// (the sharing instance for ray buffers carries six more subtractions per box and the rays' own origins: 63 registers and spills at eight waves)
__global__ __launch_bounds__(TRACE_BLOCK, (SHARE ? (PRIMARY ? P4_MIN_WAVES : 6) : P4_MIN_WAVES_WHOLE)) void k_trace_packet4(TraceArgs a)
{
    extern __shared__ uint32_t s_dynamic[];              // three stacks of PACKET_STACK words per wave of the workgroup (launch_tagged sizes it)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wave_gid = blockIdx.x * (blockDim.x >> 6) + wave;
    uint32_t* st_node = s_dynamic + wave * (3u * PACKET_STACK); uint32_t* st_lo = st_node + PACKET_STACK; uint32_t* st_hi = st_lo + PACKET_STACK;
    uint32_t* spill = nullptr;                           // taken on the first overflow of the LDS stack
    uint32_t spill_slot = 0xFFFFFFFFu;
    // The first PACKET_VSTACK entries of the (node, lane mask) stack live in LANES of three vector registers: entry i = lane i.
    // A push is three v_writelane (scalar value -> lane sp), a pop three v_readlane straight into scalar registers: no LDS
    // round trip, no single-lane region around the writes, no v_readfirstlane -- and a wave that shares its stack writes it out
    // with one store per lane.  Deeper entries go to LDS and then to the global overflow areas as before.
    uint32_t vs_node = 0u, vs_lo = 0u, vs_hi = 0u;

#if RTMI_P4_VARIANT == 3
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same clock on every XCD
#endif
    // the arguments the steps of the walk use (see `registers` above)
    const QuadNode* const quad_general = a.quad;
    const uint32_t octant_records = a.quad_octant_records;
    const TriCam* const tricam = a.tricam;
    const TriRec* const trirec = a.trirec;               // ray buffers (PRIMARY = 0): the triangles as they are, every lane its own origin
    const uint32_t budget = a.defer_budget;
    const uint32_t n_block_waves = (a.total_work + 63u) >> 6;
    const bool helper = wave_gid >= n_block_waves;       // SHARE only: the grid ends with waves that have no block of their own
    // one wave per 8x8-pixel block: the hardware dispatcher is the work queue
    uint32_t job_w0 = wave_gid << 6, job_node = 0u;
    if (!helper && a.block_order != nullptr) job_w0 = a.block_order[__builtin_amdgcn_readfirstlane((int)wave_gid)] << 6;      // (uniform index: a scalar load)
    unsigned long long job_mask = 0ull;
    bool is_task = false, have = !helper;
#if RTMI_P4_VARIANT == 3
    uint32_t dbg_jobs = 0;
#endif
    for (;;) {
    if (have) {
#if RTMI_P4_VARIANT == 3
        const unsigned long long job_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t w = job_w0 + lane;
        RayState r;
        r.o = f3{ 0, 0, 0 }; r.d = f3{ 0, 0, 1 }; r.inv = f3{ 0, 0, 1 }; r.best_t = FLT_MAX; r.best_leaf = NO_LEAF; r.best_pos = 0u;
        r.work = w; r.out_index = 0; r.pixel = 0;
        bool ok;
        uint64_t todo;                                   // the job's lanes that have not walked yet
        uint32_t node0;                                  // where they start
        uint32_t my_octant;                              // bit k: 1 / direction negative on axis k
        bool split = SHARE && is_task;                   // results go to the key buffer
        {
            // (without sharing a wave has this one job: what it reads of the arguments here is dead before the walk begins, and
            //  scalar loads of the by-value copy are back long before the flat loads of rare_args)
            const TraceArgs& A = SHARE ? rare_args<1>() : a;
            bool sx = false, sy = false, sz = false;
            ok = w < A.total_work && load_ray<PRIMARY>(A, w, r, sx, sy, sz);
            my_octant = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
            if (!SHARE || !is_task) {
                if (PRIMARY && A.compact && !ok && w < A.total_work) {
                    uint32_t x, y, slot;
                    work_to_pixel_block(A, w, x, y, slot);
                    float4* o = reinterpret_cast<float4*>(A.out + slot);
                    o[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    o[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (FUSE) store_colour(A, slot, 0u, 0.0f, NO_LEAF);
                }
                // The four-wide records skip the boxes of a node's inner children, which is sound on the search tree only (the
                // host launches this kernel only when there is one).  A ray that must walk the reference tree box by box
                // (ray_root: a slab product could be 0 * inf -- the camera on a box plane and a zero direction component) is
                // handed, whole, to the wave-per-ray kernel, which walks binary records.
                // On the per-octant records a 0 * inf on the X axis is harmless ("the first axis"): in the
                // reference's staged test (AABB.h:126-139) a NaN x product only ever stands on the side of a comparison that makes
                // it false and is replaced by the y value in the ternary max / min that follow -- exactly what -inf (near) / +inf
                // (far) would do, which is also what the products of every other box on that axis are; the test is then monotone
                // under nesting again and the search-tree argument holds.  v_max3 / v_min3 ignore a NaN operand the same way
                // (child_mask_octant).  A NaN on y or z makes the staged test forget earlier constraints -- the result depends on
                // the reference's own hierarchy -- so only those rays go to the reference tree.  (The general records' min / max of
                // the two products would turn NaN into the wrong infinity: without the copies every plane axis counts.)
                const uint32_t plane_bits = PRIMARY ? (OCT ? camera_plane_bits(A) & ~1u : camera_plane_bits(A)) : 0u;
                const uint32_t start_elsewhere = packet_hand_over<PRIMARY>(A, r, plane_bits);
                const bool on_ref = ok && start_elsewhere != RT_SENTINEL;
                const uint64_t m_ref = __ballot(on_ref);
                if (OCT && __ballot(ok && ((A.octant_mask >> my_octant) & 1u) == 0u) != 0ull) TRACE_FAIL((gu32*)&A.ctrl->error, 7);   // frame invalid
                if (m_ref != 0ull) {
                    if (A.no_unsafe_rays) TRACE_FAIL((gu32*)&A.ctrl->error, 6);       // nobody will walk them: frame invalid
                    uint32_t base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add((gu32*)&A.ctrl->long_count, (uint32_t)__popcll(m_ref), RLX_AGENT);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (on_ref) {
                        const uint32_t idx = base + lane_rank(m_ref);
                        if (idx < A.long_cap) {
                            uint32_t* rec = A.long_list + (uint64_t)idx * LONG_REC_WORDS;
                            rec[LR_WORK] = w; rec[LR_CUR] = start_elsewhere; rec[LR_SP] = 0u; rec[LR_BEST_LEAF] = NO_LEAF;
                            rec[LR_BEST_T] = __float_as_uint(FLT_MAX); rec[LR_EXTRA] = RT_SENTINEL; rec[LR_PENDING] = NO_LEAF; rec[LR_PARTIAL] = 0u;
                        } else {
                            TRACE_FAIL((gu32*)&A.ctrl->error, 1);          // nowhere to put the ray: frame invalid
                        }
                    }
                    ok = ok && !on_ref;                                                         // their contacts are written there
                }
                todo = __builtin_amdgcn_ballot_w64(ok);
                node0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.search_root);
            } else {
                todo = job_mask;
                node0 = job_node;
            }
        }
#if RTMI_P4_VARIANT == 2
        unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
#endif
#if RTMI_P4_VARIANT >= 2
        uint32_t nsteps = 0;
#endif
        uint32_t cost_steps = 0;
        // OCT: the lanes walk in groups of one direction octant each, on that octant's copy of the records (near / far planes in
        // place, child_mask_octant).  All but the blocks on the image's centre lines are one group; a task's lanes come out
        // of one group and are one again.  Without the copies: one group on the general records.
        if (cal_expired<CAL>(a)) todo = 0ull;          // (the wave still goes through its exit: counters of the sharing protocol)
        uint32_t cal_tick = 0;
        while (todo != 0ull) {
        uint64_t in = todo;                              // the lanes that walk the current node
        const QuadNode* quad = quad_general;
        if (OCT) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)my_octant, (int)__builtin_ctzll(todo));
            in = todo & __builtin_amdgcn_ballot_w64(my_octant == o);
            quad = quad_general + (uint64_t)octant_records * (1u + o);
        }
        todo &= ~in;
        uint32_t node = node0;
        // steps until the next sharing: the first time after `defer_budget` steps, then after half as many, ... down to
        // SHARE_MIN_STEPS; a task starts at half.  Only waves that run long come to share eagerly, and towards the end of a
        // launch only such waves are left -- which is when other waves have nothing to do
        uint32_t sp = 0, steps = 0, share_after = (SHARE && is_task) ? (budget >> 1 > SHARE_MIN_STEPS ? budget >> 1 : SHARE_MIN_STEPS) : budget;
        bool walking = true;
        while (walking) {
            P4_STAMP(T0);
            if (CAL == 1 && (++cal_tick & 31u) == 0u && cal_expired<CAL>(a)) { todo = 0ull; break; }
            if (SHARE && ++steps > share_after) {
                steps = 0;
                share_after = share_after >> 1 > SHARE_MIN_STEPS ? share_after >> 1 : SHARE_MIN_STEPS;
                if (sp != 0u && sp <= PACKET_VSTACK) {
                    // the whole stack becomes tasks: lane i holds entry i and offers it to shard home + i
                    // (values read through A come back in vector registers: what decides wave-uniform state is made scalar again)
                    const TraceArgs& A = rare_args<2>();
                    const uint32_t long_cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.long_cap);
                    const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.task_cap);
                    bool can = true;
                    const bool first_share = !split;
                    if (!split) {
                        uint32_t si = 0;
                        if (lane == 0) si = __hip_atomic_fetch_add((gu32*)&A.ctrl->split_tiles, 1u, RLX_AGENT);
                        si = (uint32_t)__builtin_amdgcn_readfirstlane((int)si);
                        if (si < long_cap) {
                            if (lane == 0) __hip_atomic_store((gu32*)(A.split_list + si), job_w0, RLX_AGENT);
                            if (w < A.total_work) __hip_atomic_store((gu64*)A.keys + w, KEY_NONE, RLX_AGENT);
                            split = true;
                        } else can = false;                           // no slot in the list of shared blocks: this one stays whole
                    }
                    if (can) {
                        // SHARE_BOTTOM_HALF: only the older half of the entries (the bottom of the stack: sub-trees hanging higher
                        // up, the big ones) is offered; the younger, small ones are walked here anyway before the next sharing
                        const bool mine = lane < (SHARE_BOTTOM_HALF == 1 ? (sp + 1u) / 2u : SHARE_BOTTOM_HALF == 0 ? sp : (sp < (uint32_t)SHARE_BOTTOM_HALF - 1u ? sp : (uint32_t)SHARE_BOTTOM_HALF - 1u));
                        const uint32_t shard = (wave_gid + lane) & (TASK_SHARDS - 1u);
                        uint32_t slot = 0xFFFFFFFFu;
                        if (mine) {
                            slot = __hip_atomic_fetch_add((gu32*)&A.task_shards[shard].published, 1u, RLX_AGENT);
                            if (slot >= cap) slot = 0xFFFFFFFFu;          // this queue is full: the entry stays here
                        }
                        const bool sent = mine && slot != 0xFFFFFFFFu;
                        uint32_t* e = reinterpret_cast<uint32_t*>(A.task_shards + TASK_SHARDS) + ((uint64_t)shard * cap + (sent ? slot : 0u)) * 4u;
                        // keys and split-list slot of a block shared for the first time must be there before its first task is
                        if (first_share) stores_done();
                        // one 16-byte store per entry (word 0, the block's first work item + 1, marks it written; the taker reads
                        // the entry with one 16-byte load)
                        if (sent) store16_device(e, job_w0 + 1u, vs_node, vs_lo, vs_hi);
                        // entries that stay (not offered, or their queue was full) move down (through the LDS part of the stack, empty here)
                        const uint64_t mk = __ballot(lane < sp && !sent);
                        if (mk != 0ull) {
                            if (lane < sp && !sent) { const uint32_t k = lane_rank(mk); st_node[k] = vs_node; st_lo[k] = vs_lo; st_hi[k] = vs_hi; }
                            __builtin_amdgcn_wave_barrier();
                            if (lane < (uint32_t)__popcll(mk)) { vs_node = st_node[lane]; vs_lo = st_lo[lane]; vs_hi = st_hi[lane]; }
                            __builtin_amdgcn_wave_barrier();
                        }
                        sp = (uint32_t)__popcll(mk);
                    }
                }
            }
            const uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
            const float4* p = reinterpret_cast<const float4*>(quad + un);
            const float4 q0 = ld_const4(p), q1 = ld_const4(p + 1), q2 = ld_const4(p + 2), q3 = ld_const4(p + 3);
            const float4 q4 = ld_const4(p + 4), q5 = ld_const4(p + 5), q6 = ld_const4(p + 6), q7 = ld_const4(p + 7);
            const uint32_t id0 = __float_as_uint(q0.w), id1 = __float_as_uint(q1.w), id2 = __float_as_uint(q2.w), id3 = __float_as_uint(q3.w);
            P4_KEEP("s"(id0), "s"(id1), "s"(id2), "s"(id3), "s"(q4.w), "s"(q7.x), "s"(q5.x), "s"(q3.x), "s"(q1.x));
            P4_STAMP(T1);
            P4_PIN(r.inv.x);
            // all 64 lanes evaluate all four boxes (a vector instruction costs the same for one lane as for 64; a divergent `if`
            // costs an exec-mask region, i.e. scalar instructions and a branch): lanes that do not take part are masked afterwards
            // (an unused slot's camera-relative box is empty and unmarked, k_prepare_camera: no test of its id here)
            uint64_t m[4];
            if (OCT) {
                // (round 3, tried and not kept: the set `in` and the threshold compare folded into ONE compare per slot -- lanes that do not
                //  walk the node, or whose near is not beyond the threshold, compared as NaN: one scalar AND less, one vector select more
                //  per slot -- and, separately, the leaf slots ordered first so that one test of slot 0 replaces the four leaf checks:
                //  neither moved an isolated frame nor the five-stream step (21 670 / 21 560-21 620 Mrays/s in alternating runs).  At
                //  full occupancy the kernel is bound by the latency of a step's dependent chain times the eight waves a SIMD holds
                //  (8 192 waves x ~1 step per microsecond = the 3.5 M steps of a bench step in 0.43 ms), not by instruction issue.)
                m[0] = in & child_mask_octant(q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r, q4.w);
                m[1] = in & child_mask_octant(q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r, q5.w);
                m[2] = in & child_mask_octant(q4.x, q4.y, q4.z, q5.x, q5.y, q5.z, r, q6.w);
                m[3] = in & child_mask_octant(q6.x, q6.y, q6.z, q7.x, q7.y, q7.z, r, q7.w);
            } else if (PRIMARY) {
                m[0] = in & child_mask_delta(q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r, q4.w);
                m[1] = in & child_mask_delta(q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r, q5.w);
                m[2] = in & child_mask_delta(q4.x, q4.y, q4.z, q5.x, q5.y, q5.z, r, q6.w);
                m[3] = in & child_mask_delta(q6.x, q6.y, q6.z, q7.x, q7.y, q7.z, r, q7.w);
            } else {
                // a ray buffer's packet on the records as built (k_finish): origins differ from lane to lane
                m[0] = in & child_mask_general(q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r);
                m[1] = in & child_mask_general(q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r);
                m[2] = in & child_mask_general(q4.x, q4.y, q4.z, q5.x, q5.y, q5.z, r);
                m[3] = in & child_mask_general(q6.x, q6.y, q6.z, q7.x, q7.y, q7.z, r);
            }
            const uint32_t ids[4] = { id0, id1, id2, id3 };
            P4_KEEP("s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]));
            P4_STAMP(T2);
            // leaves are tested on the spot by the lanes that reached them (an unused slot counts as a leaf nobody reaches)
            // (round 3, measured with the tight search tree, where a frame lasts as long as its heaviest block: without the triangle tests
            //  ManFigure's walk takes 0.056 instead of 0.092 ms and Tenryuu's 0.089 instead of 0.175 (-DRTMI_EXP_NO_LEAF); with the records
            //  fetched but nothing computed 0.067 / 0.126 (-DRTMI_EXP_LEAF_LOAD_ONLY): the tests' arithmetic is 27-28 % of the critical
            //  path, their scalar-load round trips 12-21 %.  Tried: the record of a reached leaf fetched by LDS-DMA (global_load_lds_dwordx4,
            //  lanes 0-3, checked alone in tools/micro/glds_leaf.hip) and its test run one step later, behind the issue of the next record's
            //  loads -- a leaf's result decides nothing about the walk.  Bit-exact, but no faster (Tenryuu 0.140 vs 0.140, ManFigure 0.106
            //  vs 0.106): the compiler waits for the first of the four record loads before it issues the rest once the deferred tests sit
            //  between the loads and their use.  Not kept.)
            // (round 3: ONE branch around the whole section -- leaf masks by select, the slots' masks cleared by select -- instead of these
            //  four mostly not-taken ones: ManFigure 0.270 -> 0.361 ms, Gardenlight 0.063 -> 0.071: the sixteen scalar selects every step
            //  pays cost more than the branches it skips; this kernel's time follows its scalar instruction count)
            if ((id0 & RT_LEAF_FLAG) != 0u) { if (m[0] != 0ull) { if (PRIMARY) leaf_test_camera(tricam, id0 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[0])); else leaf_test_record(trirec, id0 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[0])); } m[0] = 0ull; }
            if ((id1 & RT_LEAF_FLAG) != 0u) { if (m[1] != 0ull) { if (PRIMARY) leaf_test_camera(tricam, id1 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[1])); else leaf_test_record(trirec, id1 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[1])); } m[1] = 0ull; }
            if ((id2 & RT_LEAF_FLAG) != 0u) { if (m[2] != 0ull) { if (PRIMARY) leaf_test_camera(tricam, id2 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[2])); else leaf_test_record(trirec, id2 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[2])); } m[2] = 0ull; }
            if ((id3 & RT_LEAF_FLAG) != 0u) { if (m[3] != 0ull) { if (PRIMARY) leaf_test_camera(tricam, id3 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[3])); else leaf_test_record(trirec, id3 & ~RT_LEAF_FLAG, r, __builtin_amdgcn_inverse_ballot_w64(m[3])); } m[3] = 0ull; }
            P4_KEEP("v"(r.best_t), "v"(r.best_leaf), "v"(r.best_pos));
            P4_STAMP(T3);
            // the LAST wanted inner child is walked next, the ones before it wait on the stack with their lanes: a slot that is
            // wanted first sends the one held so far to the stack and is held itself (the order in which sub-trees are walked
            // does not matter: every reached leaf is tested, the hit is the lexicographic minimum over all of them)
            uint32_t held_node = 0u; uint64_t held = 0ull;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (m[k] == 0ull) continue;
                if (held != 0ull) {
                    if (sp < PACKET_VSTACK) {
                        lane_write3(vs_node, vs_lo, vs_hi, held_node, (uint32_t)held, (uint32_t)(held >> 32), sp);
                    } else {
                        const uint32_t ls = sp - PACKET_VSTACK;
                        if (ls < PACKET_STACK) {
                            if (lane == 0) { st_node[ls] = held_node; st_lo[ls] = (uint32_t)held; st_hi[ls] = (uint32_t)(held >> 32); }
                        } else {
                            const TraceArgs& A = rare_args<7>();
                            if (A.packet_spill && ls - PACKET_STACK < A.packet_spill_entries && (spill || (spill = spill_acquire(A, wave_gid, lane, spill_slot)))) {
                                if (lane == 0) { uint32_t* e = spill + (uint64_t)(ls - PACKET_STACK) * 3u; e[0] = held_node; e[1] = (uint32_t)held; e[2] = (uint32_t)(held >> 32); }
                            } else {
                                if (lane == 0) TRACE_FAIL((gu32*)&A.ctrl->error, 3);     // frame invalid
                            }
                        }
                    }
                    sp++;
                }
                held_node = ids[k]; held = m[k];
            }
            if (held != 0ull) {
                node = held_node;
                in = held;
            } else if (sp == 0u) walking = false;
            else {
                sp--;
                uint32_t nn, lo, hi;
                if (sp < PACKET_VSTACK) {
                    nn = (uint32_t)__builtin_amdgcn_readlane((int)vs_node, (int)sp);
                    lo = (uint32_t)__builtin_amdgcn_readlane((int)vs_lo, (int)sp);
                    hi = (uint32_t)__builtin_amdgcn_readlane((int)vs_hi, (int)sp);
                } else {
                    const uint32_t ls = sp - PACKET_VSTACK;
                    if (ls < PACKET_STACK) { nn = st_node[ls]; lo = st_lo[ls]; hi = st_hi[ls]; }
                    else if (spill) { const uint32_t* e = spill + (uint64_t)(ls - PACKET_STACK) * 3u; nn = e[0]; lo = e[1]; hi = e[2]; }
                    else { nn = 0u; lo = 0u; hi = 0u; }                  // the push failed (error word set): an entry nobody wants
                    nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)nn);
                    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
                    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
                }
                node = nn;
                in = ((unsigned long long)hi << 32) | lo;
            }
#if RTMI_P4_VARIANT == 2
            P4_KEEP("s"(node), "s"(in));
            P4_STAMP(T4);
            tA += T1 - T0; tB += T2 - T1; tC += T3 - T2; tD += T4 - T3;
#endif
#if RTMI_P4_VARIANT >= 2
            nsteps++;
#endif
            cost_steps++;
        }
        }                                                // next octant group
        // (a fused frame's shading arguments -- scene pointers, counts, the whole camera -- would otherwise live in scalar registers
        //  through the walk and be spilled into vector lanes at the top of every wave: they are read here, where they are used)
        const TraceArgs& A = (SHARE || FUSE) ? rare_args<4>() : a;
        // what this block cost, for the start order of later frames (launch_block_order): its own steps; a block that shared sub-trees
        // ranks before every block that did not
        if (A.block_cost != nullptr && lane == 0 && !(SHARE && is_task)) A.block_cost[job_w0 >> 6] = cost_steps + (split ? 0x10000u : 0u);
#if RTMI_P4_VARIANT == 2
        if (lane == 0) {
            __hip_atomic_fetch_add((gu32*)&A.ctrl->long_steps, (uint32_t)(tA >> 6), RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)&A.ctrl->long_steps_max, (uint32_t)(tB >> 6), RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)&A.ctrl->long_nodes, (uint32_t)(tC >> 6), RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)&A.ctrl->spill_taken, (uint32_t)(tD >> 6), RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)&A.ctrl->split_count, nsteps, RLX_AGENT);
        }
#elif RTMI_P4_VARIANT == 3
        if (lane == 0 && !SHARE) { A.keys[3u * wave_gid] = wave_t0; A.keys[3u * wave_gid + 1u] = __builtin_amdgcn_s_memrealtime(); A.keys[3u * wave_gid + 2u] = nsteps; }
        if (lane == 0 && SHARE && A.dbg) {
            unsigned long long* d = A.dbg + ((uint64_t)wave_gid * P4_DBG_JOBS + (dbg_jobs < P4_DBG_JOBS - 1u ? dbg_jobs : P4_DBG_JOBS - 1u)) * 3u;
            d[0] = dbg_jobs == 0u ? wave_t0 : job_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = ((unsigned long long)nsteps << 1) | (is_task ? 1u : 0u);
        }
        dbg_jobs++;
#endif
        if (!split) {
            if (ok) {
                store_contact(A, r.out_index, r.pixel, r.best_t, r.best_leaf);
                if (FUSE) store_colour(A, r.out_index, r.pixel, r.best_t, r.best_leaf);
            }
        } else if (r.best_leaf != NO_LEAF) {
            __hip_atomic_fetch_min((gu64*)A.keys + w, ((unsigned long long)__float_as_uint(r.best_t) << 32) | r.best_pos, RLX_AGENT);
        }
        if (SHARE) {
            // the helpers read keys and the split list after they have seen these counters
            if (split) stores_done();
            TaskShard* mine = A.task_shards + (wave_gid & (TASK_SHARDS - 1u));
            if (lane == 0) __hip_atomic_fetch_add((gu32*)(is_task ? &mine->tasks_done : &mine->waves_done), 1u, RLX_AGENT);
        }
    }
    if (!SHARE) break;
    // ---------------- the next job: a task, if one is there (helpers wait for one until nothing is running) ----------------
    const TraceArgs& A = rare_args<5>();
    TaskShard* const shards = A.task_shards;
    uint32_t got_shard = 0xFFFFFFFFu, got_slot = 0u;
    if (helper && wave_gid == n_block_waves) {
        // the watching wave: lane l reads shards l, l + 64, ...
        const uint32_t cap = A.task_cap;
        for (;;) {
            uint32_t wd = 0, td = 0, pub = 0;
#pragma unroll
            for (uint32_t k = 0; k < TASK_SHARDS / 64u; k++) wd += __hip_atomic_load((gu32*)&shards[lane + 64u * k].waves_done, RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wd) :: "memory");
#pragma unroll
            for (uint32_t k = 0; k < TASK_SHARDS / 64u; k++) td += __hip_atomic_load((gu32*)&shards[lane + 64u * k].tasks_done, RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(td) :: "memory");
#pragma unroll
            for (uint32_t k = 0; k < TASK_SHARDS / 64u; k++) { const uint32_t c = __hip_atomic_load((gu32*)&shards[lane + 64u * k].published, RLX_AGENT); pub += c < cap ? c : cap; }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pub) :: "memory");
            const uint32_t published = wave_sum(pub);
            if (wave_sum(wd) == n_block_waves && wave_sum(td) == published) {
                if (lane == 0) A.ctrl->shared_tasks = published;          // diagnostic
#pragma unroll
                for (uint32_t k = 0; k < TASK_SHARDS / 64u; k++) __hip_atomic_store((gu32*)&shards[lane + 64u * k].finished, 1u, RLX_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    } else {
        for (uint32_t round = 0;; round++) {
            // a block wave looks at ONE shard (a device-scope load of one address costs ~60 ns at the memory side whoever issues
            // it: eight shards per wave were 0.25 ms per frame), a helper at two per round, all the shards in turn; helpers read
            // the shard's `finished` word along with it
            const uint32_t look = helper ? (wave_gid * 2u + 2u * round + lane) & (TASK_SHARDS - 1u) : (wave_gid * 7u + round) & (TASK_SHARDS - 1u);
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (lane < (helper ? 2u : 1u)) {
                const u32x4 v = load16_device(&shards[look].taken);
                const uint32_t cap = A.task_cap;
                q = make_uint4(v.x, v.y < cap ? v.y : cap, v.z, 0u);
            }
            const uint64_t avail = __ballot(q.x < q.y);
            if (avail != 0ull) {
                const uint32_t pick = (uint32_t)__builtin_ctzll(avail);
                uint32_t ok_slot = 0xFFFFFFFFu;
                if (lane == pick) {
                    uint32_t expect = q.x;
                    if (__hip_atomic_compare_exchange_strong((gu32*)&shards[look].taken, &expect, q.x + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        ok_slot = q.x;
                }
                ok_slot = (uint32_t)__builtin_amdgcn_readlane((int)ok_slot, (int)pick);
                if (ok_slot != 0xFFFFFFFFu) { got_shard = (uint32_t)__builtin_amdgcn_readlane((int)look, (int)pick); got_slot = ok_slot; break; }
                if (!helper && round >= 3u) break;            // lost the race a few times: others are on it
                // lost the race: do not come back at once with everybody else who lost it
                __builtin_amdgcn_s_sleep(8);
                if ((wave_gid + round) & 1u) __builtin_amdgcn_s_sleep(8);
                continue;
            }
            if (!helper) break;
            if (__ballot(q.z != 0u) != 0ull) break;           // the watching wave has seen everything finished
            __builtin_amdgcn_s_sleep(96);
        }
    }
    if (got_shard == 0xFFFFFFFFu) break;
    {
        // the publisher reserved the entry before it wrote it: wait for the valid mark (a few instructions away)
        const uint32_t* e = reinterpret_cast<const uint32_t*>(shards + TASK_SHARDS) + ((uint64_t)got_shard * A.task_cap + got_slot) * 4u;
        u32x4 t;
        do { t = load16_device(e); } while (t.x == 0u);
        job_w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x) - 1u;
        job_node = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y);
        job_mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)t.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)t.z);
        is_task = true; have = true;
    }
    }
    const TraceArgs& A = SHARE ? rare_args<6>() : a;          // (without sharing only the overflow bitmap is read here)
    if (SHARE && helper) {
        // everything has finished: the shared blocks' contacts come out of the key buffer
        uint32_t n_split = __hip_atomic_load((gu32*)&A.ctrl->split_tiles, RLX_AGENT);
        if (n_split > A.long_cap) n_split = A.long_cap;
        const uint32_t n_helpers = gridDim.x * (blockDim.x >> 6) - n_block_waves;
        const uint32_t plane_bits = PRIMARY ? (OCT ? camera_plane_bits(A) & ~1u : camera_plane_bits(A)) : 0u;     // as where the blocks start
        for (uint32_t i = wave_gid - n_block_waves; i < n_split; i += n_helpers) {
            const uint32_t w = __hip_atomic_load((gu32*)(A.split_list + i), RLX_AGENT) + lane;
            RayState r;
            bool sx, sy, sz;
            r.out_index = 0; r.pixel = 0;
            if (w < A.total_work && load_ray<PRIMARY>(A, w, r, sx, sy, sz) && packet_hand_over<PRIMARY>(A, r, plane_bits) == RT_SENTINEL) {
                const unsigned long long key = __hip_atomic_load((gu64*)A.keys + w, RLX_AGENT);
                const bool hit = key != KEY_NONE;
                const float t = hit ? __uint_as_float((uint32_t)(key >> 32)) : FLT_MAX;
                const uint32_t leaf = hit ? A.pairs[(uint32_t)key].y : NO_LEAF;
                store_contact(A, r.out_index, r.pixel, t, leaf);
                if (FUSE) store_colour(A, r.out_index, r.pixel, t, leaf);
            }
        }
    }
    spill_release(A, lane, spill_slot);
}

// ------------------------------------------------------------------------------------------------
// phase 1: up to LONG_BATCH deferred rays per wave, 64 lanes walk their sub-trees together
// ------------------------------------------------------------------------------------------------
// One deferred ray keeps 16-37 of 64 lanes busy per pool step (ClumsyDragon / Tenryuu, RTMI_TRACE_DIAG=1) and the kernel is
// bound by instruction issue, so a wave takes several records at once: pool entries carry the ray's slot in their top
// bits, the rays live in LDS (two 16-byte reads per lane and step instead of registers), and hits meet in a 64-bit
// (t bits, position) key per slot through an LDS atomic min -- the same lexicographic minimum the one-ray version reduced
// at the end, so results do not depend on which rays share a wave.
#define LONG_BATCH 4u                          // most rays per wave; the kernel takes 1, 2 or 4 (see `batch`)
#define LONG_SLOT_SHIFT 28u                     // ids below 2^28: rt_bvh_construct refuses more than 2^27 triangles
#define LONG_ID_MASK 0x0FFFFFFFu
#define LONG_LIVE 1u
#define LONG_PARTIAL 2u

template <int PRIMARY, int CAL>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_long(TraceArgs a)
{
    __shared__ uint32_t s_inner[TRACE_WAVES][POOL_INNER];
    __shared__ uint32_t s_leaf[TRACE_WAVES][POOL_LEAF];
    // per slot: (o.xyz, signs) (d.xyz, work) (inv.xyz, flags) (out_index, pixel, -, -)
    __shared__ float4 s_ray[TRACE_WAVES][LONG_BATCH][4];
    __shared__ unsigned long long s_key[TRACE_WAVES][LONG_BATCH];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* inner = s_inner[wave];
    uint32_t* leafp = s_leaf[wave];
    float4 (*ray)[4] = s_ray[wave];
    unsigned long long* keys = s_key[wave];
    const uint32_t wave_gid = blockIdx.x * TRACE_WAVES + wave;
    gu32* error_word = (gu32*)&a.ctrl->error;
    uint32_t count = a.ctrl->long_count;                  // written by the previous kernel
    if (count > a.long_cap) count = a.long_cap;
    // rays per wave: sharing a wave pays only while there are more records than resident waves (a frame with few deferred
    // rays is latency bound, and four rays in one wave are then four times the latency: ClumsyDragon 0.11 -> 0.15 ms)
    // (Tenryuu, 63 k / 110 k records on ~6 k resident waves: 0.47 / 0.73 ms alone, 0.36 / 0.50 in pairs, 0.37 / 0.46 in fours,
    // 0.48 / 0.56 in eights)
    const uint32_t batch = a.long_batch ? a.long_batch : count >= 4u * a.n_waves ? 4u : count >= a.n_waves ? 2u : 1u;
    const uint32_t n_batches = (count + batch - 1u) / batch;

    // deferred rays differ in length by an order of magnitude: the first batch of a wave is static (its own index),
    // further batches are pulled from sharded counters
    uint32_t bi = wave_gid;
    uint32_t shard = wave_gid % WORK_SHARDS, shards_tried = 0;
    for (;;) {
        if (bi >= n_batches) break;
        if (cal_expired<CAL>(a)) break;
        // ---------------- set the batch up: lane s < batch prepares slot s ----------------
        if (lane < batch) {
            const uint32_t i = bi * batch + lane;
            bool live = i < count;
            const uint32_t* rec = a.long_list + (uint64_t)(live ? i : 0u) * LONG_REC_WORDS;
            uint32_t work = live ? rec[LR_WORK] : 0xFFFFFFFFu;
            if (work == 0xFFFFFFFFu) live = false;       // reserved by a block that found the list full
            if (live) {
                // a record that does not describe a task of this launch must never be walked (it would read anywhere):
                // report it through the error word and skip it
                const uint32_t n_leaves = a.n_leaves;
                auto bad_id = [&](uint32_t id) {
                    return id != RT_SENTINEL && ((id & RT_LEAF_FLAG) ? (id & ~RT_LEAF_FLAG) >= n_leaves : id >= a.n_records);
                };
                const bool bad = work >= a.total_work || rec[LR_SP] > LDS_STACK || bad_id(rec[LR_CUR]) || bad_id(rec[LR_EXTRA]) ||
                                 (rec[LR_PENDING] != NO_LEAF && rec[LR_PENDING] >= n_leaves) ||
                                 (rec[LR_BEST_LEAF] != NO_LEAF && rec[LR_BEST_LEAF] >= n_leaves);
                if (bad) { TRACE_FAIL(error_word, 4); live = false; }
            }
            RayState r;
            bool sx, sy, sz;
            load_ray<PRIMARY, false>(a, live ? work : 0u, r, sx, sy, sz);          // (each lane another record's ray)
            const uint32_t signs = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
            const uint32_t flags = (live ? LONG_LIVE : 0u) | (live && rec[LR_PARTIAL] != 0u ? LONG_PARTIAL : 0u);
            ray[lane][0] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(signs));
            ray[lane][1] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(work));
            ray[lane][2] = make_float4(r.inv.x, r.inv.y, r.inv.z, __uint_as_float(flags));
            ray[lane][3] = make_float4(__uint_as_float(r.out_index), __uint_as_float(r.pixel), 0.0f, 0.0f);
            unsigned long long key = KEY_NONE;            // the best hit found in phase 0
            if (live && rec[LR_BEST_LEAF] != NO_LEAF) key = ((unsigned long long)rec[LR_BEST_T] << 32) | a.pos[rec[LR_BEST_LEAF]];
            keys[lane] = key;
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- seed the pools, slot by slot: lanes 0..sp-1 take the stack, lanes 62/63 the current node and the extra child
        uint32_t n_inner = 0, n_leaf = 0;
        for (uint32_t sl = 0; sl < batch; sl++) {
            if ((__float_as_uint(ray[sl][2].w) & LONG_LIVE) == 0u) continue;                 // wave-uniform
            const uint32_t* rec = a.long_list + (uint64_t)(bi * batch + sl) * LONG_REC_WORDS;
            // header words are read by every lane (same address: one broadcast fetch of the 128-byte record)
            const uint32_t rcur = rec[LR_CUR], rsp = rec[LR_SP], rextra = rec[LR_EXTRA], rpending = rec[LR_PENDING];
            uint32_t seed = RT_SENTINEL;
            if (lane < rsp) seed = rec[LR_STACK + lane];
            if (lane == 62) seed = rcur;
            if (lane == 63) seed = rextra;
            const bool is_in = seed != RT_SENTINEL && (seed & RT_LEAF_FLAG) == 0u;
            const bool is_lf = seed != RT_SENTINEL && (seed & RT_LEAF_FLAG) != 0u;
            const uint64_t mi = __ballot(is_in), ml = __ballot(is_lf);
            const uint32_t tag = sl << LONG_SLOT_SHIFT;
            if (is_in) inner[n_inner + lane_rank(mi)] = seed | tag;
            if (is_lf) leafp[n_leaf + lane_rank(ml)] = (seed & ~RT_LEAF_FLAG) | tag;
            n_inner += (uint32_t)__popcll(mi); n_leaf += (uint32_t)__popcll(ml);
            if (rpending != NO_LEAF) { if (lane == 0) leafp[n_leaf] = rpending | tag; n_leaf++; }
        }

        uint32_t pool_steps = 0, pool_nodes = 0;
        while (n_inner != 0u || n_leaf != 0u) {
            pool_steps++;
            if (CAL == 1 && (pool_steps & 15u) == 0u && cal_expired<CAL>(a)) break;
            // ---- leaves, 64 at a time, when the pool is filling up or nothing else is left ----
            if (n_leaf >= POOL_LEAF - 128u || n_inner == 0u) {
                uint32_t k = n_leaf < 64u ? n_leaf : 64u;
                if (lane < k) {
                    // triangleIntersect, Triangle.h:66-107, as in leaf_test; the hit goes into the slot's key
                    const uint32_t e = leafp[n_leaf - 1u - lane];
                    const uint32_t sl = e >> LONG_SLOT_SHIFT, leaf = e & LONG_ID_MASK;
                    const float4 ro = ray[sl][0], rd = ray[sl][1];
                    const f3 o = f3{ ro.x, ro.y, ro.z }, d = f3{ rd.x, rd.y, rd.z };
                    float4 v0 = a.v0[leaf], E1 = a.e1[leaf], E2 = a.e2[leaf];
                    f3 vert0 = f3_from4(v0), edge1 = f3_from4(E1), edge2 = f3_from4(E2);
                    f3 pvec = f3_cross(d, edge2);
                    float det = f3_dot(edge1, pvec);
                    bool flag = fabsf(det) >= FLT_EPSILON;
                    float inv_det = 1.0f / det;
                    f3 tvec = f3_sub(o, vert0);
                    float u = f3_dot(tvec, pvec) * inv_det;
                    flag = flag & (u < 1.0f);
                    flag = flag & (u >= 0.0f);
                    f3 qvec = f3_cross(tvec, edge1);
                    float v = f3_dot(d, qvec) * inv_det;
                    flag = flag & ((u + v) <= 1.0f);
                    flag = flag & (v >= 0.0f);
                    float t = f3_dot(edge2, qvec) * inv_det;
                    // a hit replaces the best only if t < best (BVH.h:392), and the best starts at FLT_MAX
                    if (flag & (t > 0.0f) & (t < FLT_MAX)) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | a.pos[leaf];
                        if (key < keys[sl]) __hip_atomic_fetch_min(&keys[sl], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                n_leaf -= k;
                continue;
            }
            // ---- inner nodes: pop up to 64 from the top of the pool ----
            // close to the pool's capacity the step degrades to a single-node depth-first walk,
            // whose growth is bounded by the tree depth
            uint32_t k = n_inner < 64u ? n_inner : 64u;
            if (n_inner > POOL_INNER - 192u) k = 1u;
            // Spare lanes (pool thinner than the wave) run AHEAD along the chain below the top node.  Duplicate Morton codes make
            // the radix tree a chain there (node i: child A = a leaf, child B = node i+1, BVH.h:138-160,190-215), and
            // consecutive chain nodes are consecutive records.  Lane j speculatively tests record top+j; it counts
            // only if every record before it is valid-B and links to its successor (prefix AND through a ballot),
            // so exactly the reached nodes contribute -- up to 64 chain nodes per step instead of one.
            // lanes 0..k-1 take the k popped nodes (lane 0 the top one); lanes k..63 run ahead along the chain below
            // the top node, for the top node's ray: lane k+j-1 speculatively takes record top+j
            const uint32_t top_e = inner[n_inner - 1u];
            const uint32_t top = top_e & LONG_ID_MASK;
            const bool spec_lane = lane >= k;
            const uint32_t seq_j = spec_lane ? lane - k + 1u : 0u;          // position in the sequence top, top+1, ...
            const uint32_t e = spec_lane ? top_e : inner[n_inner - 1u - (lane < k ? lane : 0u)];
            const uint32_t sl = e >> LONG_SLOT_SHIFT;
            const uint32_t tag = sl << LONG_SLOT_SHIFT;
            uint32_t node = spec_lane ? top + seq_j : (e & LONG_ID_MASK);
            bool mine = spec_lane ? node < a.n_records : true;
            n_inner -= k;
            bool aIn = false, aLf = false, bIn = false, bLf = false;
            uint32_t ca = 0, cb = 0;
            bool bValid = false;
            if (mine) {
                const float4 ro = ray[sl][0], ri = ray[sl][2];
                RayState r;
                r.o = f3{ ro.x, ro.y, ro.z }; r.inv = f3{ ri.x, ri.y, ri.z };
                const uint32_t signs = __float_as_uint(ro.w);
                const bool sx = (signs & 1u) != 0u, sy = (signs & 2u) != 0u, sz = (signs & 4u) != 0u;
                const float4* p = reinterpret_cast<const float4*>(a.trav + node);
                float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                ca = __float_as_uint(q0.w); cb = __float_as_uint(q1.w);
                const uint32_t marks = __float_as_uint(q2.w);
                const bool on_search = a.search_root != 0u && node >= a.search_root;
                bool aValid = child_valid_on<PRIMARY>(on_search, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, r, sx, sy, sz, (marks & 1u) != 0u);
                bValid = child_valid_on<PRIMARY>(on_search, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, r, sx, sy, sz, (marks & 2u) != 0u);
                aIn = aValid & ((ca & RT_LEAF_FLAG) == 0u); aLf = aValid & ((ca & RT_LEAF_FLAG) != 0u);
                bIn = bValid & ((cb & RT_LEAF_FLAG) == 0u); bLf = bValid & ((cb & RT_LEAF_FLAG) != 0u);
            }
            {
                // link: this record hands over to the next record (valid child B that is the consecutive inner record)
                uint64_t lk = __ballot(mine && bValid && cb == node + 1u);
                uint64_t seq = (lk & 1ull) | (k < 64u ? ((lk >> k) << 1) : 0ull);          // links in sequence order
                uint32_t reached = (uint32_t)__ffsll((unsigned long long)~seq);           // reached sequence elements
                if (reached == 0u || reached > 65u - k) reached = 65u - k;
                pool_nodes += k + reached - 1u;
                bool in_prefix = !spec_lane || seq_j < reached;
                bool emits_b = spec_lane ? (seq_j + 1u == reached) : (lane == 0u ? reached == 1u : true);
                aIn = aIn & in_prefix; aLf = aLf & in_prefix;
                bIn = bIn & in_prefix & emits_b; bLf = bLf & in_prefix & emits_b;         // the chain's B child is emitted once, by its last reached node
            }
            uint64_t mai = __ballot(aIn), mbi = __ballot(bIn), mal = __ballot(aLf), mbl = __ballot(bLf);
            uint32_t nai = (uint32_t)__popcll(mai), nbi = (uint32_t)__popcll(mbi);
            uint32_t nal = (uint32_t)__popcll(mal), nbl = (uint32_t)__popcll(mbl);
            if (n_inner + nai + nbi > POOL_INNER || n_leaf + nal + nbl > POOL_LEAF) {
                if (lane == 0) TRACE_FAIL(error_word, 2);   // frame invalid, reported by the host
                n_inner = 0; n_leaf = 0;
                break;
            }
            if (aIn) inner[n_inner + lane_rank(mai)] = ca | tag;
            if (bIn) inner[n_inner + nai + lane_rank(mbi)] = cb | tag;
            if (aLf) leafp[n_leaf + lane_rank(mal)] = (ca & ~RT_LEAF_FLAG) | tag;
            if (bLf) leafp[n_leaf + nal + lane_rank(mbl)] = (cb & ~RT_LEAF_FLAG) | tag;
            n_inner += nai + nbi;
            n_leaf += nal + nbl;
        }

        if (a.diagnostics && lane == 0) {       // same-address atomics cost ~11 ns each: only on request (RTMI_TRACE_DIAG=1)
            __hip_atomic_fetch_add((gu32*)&a.ctrl->long_steps, pool_steps, RLX_AGENT);
            __hip_atomic_fetch_max((gu32*)&a.ctrl->long_steps_max, pool_steps, RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)&a.ctrl->long_nodes, pool_nodes, RLX_AGENT);
        }
        // ---------------- results: lane s < batch finishes slot s ----------------
        __builtin_amdgcn_wave_barrier();
        if (lane < batch) {
            const uint32_t flags = __float_as_uint(ray[lane][2].w);
            if (flags & LONG_LIVE) {
                const unsigned long long key = keys[lane];
                const uint32_t work = __float_as_uint(ray[lane][1].w);
                if (flags & LONG_PARTIAL) {
                    if (key != KEY_NONE) __hip_atomic_fetch_min((gu64*)a.keys + work, key, RLX_AGENT);   // meets the other parts in k_resolve_split
                } else {
                    const uint32_t leaf = key != KEY_NONE ? a.pairs[(uint32_t)key].y : NO_LEAF;
                    const float t = key != KEY_NONE ? __uint_as_float((uint32_t)(key >> 32)) : FLT_MAX;
                    store_contact(a, __float_as_uint(ray[lane][3].x), __float_as_uint(ray[lane][3].y), t, leaf);
                    if (PRIMARY && a.fused_rgba != nullptr) store_colour(a, __float_as_uint(ray[lane][3].x), __float_as_uint(ray[lane][3].y), t, leaf);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // next batch
        bi = 0xFFFFFFFFu;
        uint32_t dyn = n_batches > a.n_waves ? n_batches - a.n_waves : 0u;
        uint32_t per = (dyn + WORK_SHARDS - 1u) / WORK_SHARDS;
        while (bi == 0xFFFFFFFFu && shards_tried < WORK_SHARDS) {
            uint32_t lo = shard * per, hi = lo + per < dyn ? lo + per : dyn;
            uint32_t got = 0xFFFFFFFFu;
            if (lo < hi) {
                if (lane == 0) got = __hip_atomic_fetch_add((gu32*)&a.ctrl->lshard[shard].next, 1u, RLX_AGENT);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            }
            if (lo < hi && got < hi - lo) bi = a.n_waves + lo + got;
            else { shard = shard + 1u == WORK_SHARDS ? 0u : shard + 1u; shards_tried++; }
        }
        if (bi == 0xFFFFFFFFu) break;
    }
}

// split rays (packet kernel): the minimum of all parts' (t, position) keys becomes the contact; leaf = pairs[position].y
template <int PRIMARY, int CAL>
__global__ __launch_bounds__(256) void k_resolve_split(TraceArgs a)
{
    uint32_t count = a.ctrl->split_count;
    if (count > a.long_cap) count = a.long_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        uint32_t w = a.split_list[i];
        unsigned long long key = __hip_atomic_load((gu64*)a.keys + w, RLX_AGENT);
        if (key == KEY_NONE) continue;                 // no hit in any part: the block's own "no contact" stands
        uint32_t out_index = w, pixel = w;
        if (PRIMARY) {
            uint32_t x, y, slot;
            work_to_pixel(a, w, x, y, slot);
            pixel = y * a.cam.resX + x;
            out_index = a.compact ? slot : pixel;
        }
        store_contact(a, out_index, pixel, __uint_as_float((uint32_t)(key >> 32)), a.pairs[(uint32_t)key].y);
    }
}

// CAL = 1 marks the launches of the first-frame calibration: same code under different kernel names, so that
// profiles (rocprofv3 --stats) keep them apart from the frames that count
template <int PRIMARY, int CAL>
static void launch_tagged(hipStream_t s, const TraceLaunch& t, TraceArgs& a)
{
    if (a.total_work == 0) return;
    // re-initialise every shared word before every launch (cdna_hip_programming.md, Guideline 16) -- unless the launch reads
    // none: the four-wide kernel without sharing touches the control block only to hand over unsafe rays (there are none, the
    // caller says), for its sticky error word and for a diagnostic count of overflow areas (which then runs on over such frames)
    const bool four_wide = PRIMARY == 1 && t.packet && t.packet4 && t.quad && t.tricam && t.search_root;
    const bool shares = four_wide && t.task_shards && t.task_cap && a.defer_budget < (1u << 20);
    GraphNodes* const rec = t.graph_record; GraphNodes* const upd = t.graph_update;      // (update mode: arguments are set on captured nodes, nothing is enqueued)
    const bool cleared = a.cleared_by_preparation != 0u;          // (an eager frame's preparation kernel zeroed the control block and the task queues)
    if (!(four_wide && !shares && t.no_unsafe_rays) && !upd && !cleared)
        (void)hipMemsetAsync(t.ctrl, 0, offsetof(TraceCtrl, error), s);      // the error word is sticky (host-cleared)
    // First-phase kernels: one wave per 64-ray block and the hardware dispatcher as the work queue.  (A persistent grid --
    // resident waves pulling blocks from sharded counters, idle lanes refilled -- measured 25-50 % slower on every bundled
    // scene: Cube 0.157 -> 0.106 ms, ManFigure lane-per-ray 1.25 -> 0.81 ms)
    // Workgroups of the first-phase kernels: TRACE_WAVES waves (neighbouring blocks of a frame on one CU: they walk the same records),
    // or ONE wave when several frames are in flight (rt_ctx_set_frames_in_flight): the dispatcher then places single waves wherever a
    // slot frees up instead of waiting for four on one CU -- five bundled scenes in flight 0.282 -> 0.260 ms per step, an isolated
    // ManFigure frame 0.105 -> 0.110 (the LDS arrays stay sized for TRACE_WAVES: 4.6 KB per workgroup, 32 of them fit a CU)
    const uint32_t wpb = (t.waves_per_block >= 1u && t.waves_per_block <= TRACE_WAVES) ? t.waves_per_block : TRACE_WAVES;
    const uint32_t grid = ((a.total_work + 63u) / 64u + wpb - 1u) / wpb;
    // The wave-per-ray kernel does not know its work on the host: resident waves pull batches of deferred rays.  Resident
    // blocks per CU are a property of the code object, asked once (the query costs microseconds of host time).
    static int b1 = 0;
    if (b1 == 0) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, k_trace_long<PRIMARY, CAL>, TRACE_BLOCK, 0); if (b1 < 1) b1 = 1; }
    const uint32_t grid1 = (uint32_t)t.compute_units * (uint32_t)b1;
    a.n_waves = grid1 * TRACE_WAVES;                     // resident waves of the wave-per-ray kernel
    if constexpr (PRIMARY == 1) if (t.packet && t.packet4 && t.quad && t.tricam && t.search_root) {   // (primary rays only: no instance for ray buffers)
        if (t.fused_rgba) {
            // every contact of this frame is stored by the kernel below -- or, for a ray it hands over whole (a camera on a box plane), by the
            // wave-per-ray kernel behind it: each shades what it stores (one launch and one pass over the contacts less per frame)
            a.fused_rgba = (float4*)t.fused_rgba; a.fused_plane_stride = t.fused_plane_stride;
            a.fused_scene.scene = t.scene->bytes; a.fused_scene.n_lights = t.scene->n_lights; a.fused_scene.n_spheres = t.scene->n_spheres;
            a.fused_scene.n_materials = t.scene->n_materials;
            if (t.fused_done) *t.fused_done = true;
        }
        a.no_unsafe_rays = t.no_unsafe_rays ? 1u : 0u;
        if (t.ev[0]) (void)hipEventRecord(t.ev[0], s);
        if (t.task_shards && t.task_cap && a.defer_budget < (1u << 20)) {
            // with a step budget: waves share their waiting sub-trees (see the kernel); queue state and valid marks start at zero
            if (!upd && !cleared) (void)hipMemsetAsync(t.task_shards, 0, trace_task_bytes(t.task_cap), s);
            const uint32_t helpers = (t.task_helper_blocks ? t.task_helper_blocks : 256u) * TRACE_WAVES / wpb;      // 256 x TRACE_WAVES waves = one per SIMD: far fewer than wave slots
#if RTMI_P4_VARIANT == 3
            static unsigned long long* dbg = nullptr; static size_t dbg_n = 0;
            const size_t want = (size_t)(grid + helpers) * TRACE_WAVES * P4_DBG_JOBS * 3u;
            if (getenv("RTMI_P4_WAVES")) {
                if (want > dbg_n) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, want * 8u); dbg_n = want; }
                (void)hipMemsetAsync(dbg, 0, want * 8u, s);
                a.dbg = dbg;
            }
#endif
            const dim3 g(grid + helpers), blk(wpb * 64u);
            const uint32_t lds = wpb * 3u * PACKET_STACK * 4u;
            if (a.quad_octant_records) { if (a.fused_rgba) rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 1, 1, 1>, g, blk, s, a, lds); else rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 1, 1, 0>, g, blk, s, a, lds); }
            else { if (a.fused_rgba) rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 1, 0, 1>, g, blk, s, a, lds); else rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 1, 0, 0>, g, blk, s, a, lds); }
#if RTMI_P4_VARIANT == 3
            if (const char* path = getenv("RTMI_P4_WAVES")) {
                std::vector<unsigned long long> h(want);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h.data(), dbg, want * 8u, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(path, "wb")) { const unsigned long long hdr[3] = { 0xFFFFFFFFull, (unsigned long long)(grid + helpers) * TRACE_WAVES, P4_DBG_JOBS }; fwrite(hdr, 8u, 3, f); fwrite(h.data(), 8u, want, f); fclose(f); }
            }
#endif
        } else {
            const dim3 g(grid), blk(wpb * 64u);
            const uint32_t lds = wpb * 3u * PACKET_STACK * 4u;
            if (a.quad_octant_records) { if (a.fused_rgba) rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 0, 1, 1>, g, blk, s, a, lds); else rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 0, 1, 0>, g, blk, s, a, lds); }
            else { if (a.fused_rgba) rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 0, 0, 1>, g, blk, s, a, lds); else rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet4<1, CAL, 0, 0, 0>, g, blk, s, a, lds); }
#if RTMI_P4_VARIANT == 3
            if (const char* path = getenv("RTMI_P4_WAVES")) {
                const size_t n = (size_t)grid * TRACE_WAVES * 3u;
                std::vector<unsigned long long> h(n);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h.data(), t.keys, n * 8u, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8u, n, f); fclose(f); }
            }
#endif
        }
        if (t.ev[1]) (void)hipEventRecord(t.ev[1], s);
        // rays the four-wide records may not be used for (see the kernel) wait on the deferred list: normally none, and the
        // wave-per-ray kernel returns at once (left out where the caller knows that there are none: frame graphs)
        // (a smaller grid for this usually empty launch was measured: the 6.4 us are the launch, not its size)
        if (!t.no_unsafe_rays) rt_launch_node(rec, upd, RT_NODE_LONG, k_trace_long<PRIMARY, CAL>, dim3(grid1), dim3(TRACE_BLOCK), s, a);
        if (t.ev[2]) (void)hipEventRecord(t.ev[2], s);
        return;
    }
    if constexpr (PRIMARY == 0) if (t.packet && t.packet4 && t.quad && t.trirec && t.search_root) {
        // A ray buffer walked as packets: 64 consecutive rays per wave on the four-wide records as built (every lane its own origin,
        // child_mask_general), sub-trees shared between waves under a step budget as for primary frames.  Rays that may not use the
        // search tree (ray_root) are handed, whole, to the wave-per-ray kernel.
        if (t.ev[0]) (void)hipEventRecord(t.ev[0], s);
        const uint32_t lds = wpb * 3u * PACKET_STACK * 4u;
        if (t.task_shards && t.task_cap && a.defer_budget < (1u << 20)) {
            (void)hipMemsetAsync(t.task_shards, 0, trace_task_bytes(t.task_cap), s);
            const uint32_t helpers = (t.task_helper_blocks ? t.task_helper_blocks : 256u) * TRACE_WAVES / wpb;
            hipLaunchKernelGGL((k_trace_packet4<0, CAL, 1, 0, 0>), dim3(grid + helpers), dim3(wpb * 64u), lds, s, a);
        } else {
            hipLaunchKernelGGL((k_trace_packet4<0, CAL, 0, 0, 0>), dim3(grid), dim3(wpb * 64u), lds, s, a);
        }
        if (t.ev[1]) (void)hipEventRecord(t.ev[1], s);
        hipLaunchKernelGGL((k_trace_long<0, CAL>), dim3(grid1), dim3(TRACE_BLOCK), 0, s, a);
        if (t.ev[2]) (void)hipEventRecord(t.ev[2], s);
        return;
    }
    if (PRIMARY && t.packet) {
        if (t.ev[0]) (void)hipEventRecord(t.ev[0], s);
        rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace_packet<PRIMARY, CAL>, dim3((a.total_work + TRACE_BLOCK - 1u) / TRACE_BLOCK), dim3(TRACE_BLOCK), s, a);     // (always TRACE_WAVES waves)
        if (t.ev[1]) (void)hipEventRecord(t.ev[1], s);
        // without a budget the packet kernel hands nothing over: the wave-per-ray and resolve kernels would find empty lists
        if (a.defer_budget < (1u << 20)) {
            rt_launch_node(rec, upd, RT_NODE_LONG, k_trace_long<PRIMARY, CAL>, dim3(grid1), dim3(TRACE_BLOCK), s, a);
            rt_launch_node(rec, upd, RT_NODE_RESOLVE, k_resolve_split<PRIMARY, CAL>, dim3(64), dim3(256), s, a);
        }
        if (t.ev[2]) (void)hipEventRecord(t.ev[2], s);
        return;
    }
    if constexpr (PRIMARY == 1) if (t.fused_rgba) {
        // lane-per-ray + wave-per-ray: between them the two kernels store every contact of the frame exactly once (a ray finishes in
        // the first or is handed, whole, to the second), so in a fused frame each shades what it stores and k_shade is not launched
        a.fused_rgba = (float4*)t.fused_rgba; a.fused_plane_stride = t.fused_plane_stride;
        a.fused_scene.scene = t.scene->bytes; a.fused_scene.n_lights = t.scene->n_lights; a.fused_scene.n_spheres = t.scene->n_spheres;
        a.fused_scene.n_materials = t.scene->n_materials;
        if (t.fused_done) *t.fused_done = true;
    }
    if (t.ev[0]) (void)hipEventRecord(t.ev[0], s);
    rt_launch_node(rec, upd, RT_NODE_TRACE, k_trace<PRIMARY, CAL>, dim3(grid), dim3(wpb * 64u), s, a, wpb * LDS_STACK * 64u * 4u);
    if (t.ev[1]) (void)hipEventRecord(t.ev[1], s);
    // wave-per-ray phase: the number of deferred rays lives on the device; a fixed grid pulls from the list
    rt_launch_node(rec, upd, RT_NODE_LONG, k_trace_long<PRIMARY, CAL>, dim3(grid1), dim3(TRACE_BLOCK), s, a);
    if (t.ev[2]) (void)hipEventRecord(t.ev[2], s);
}

template <int PRIMARY>
static void launch(hipStream_t s, const TraceLaunch& t, TraceArgs& a)
{
    if (t.tag == 1) launch_tagged<PRIMARY, 1>(s, t, a);
    else if (t.tag == 2) launch_tagged<PRIMARY, 2>(s, t, a);
    else launch_tagged<PRIMARY, 0>(s, t, a);
}

uint32_t trace_packet_stack_entries() { return PACKET_STACK; }

static void fill_common(const TraceLaunch& t, TraceArgs& a)
{
    a.trav = t.trav; a.quad = t.search_root ? t.quad : nullptr; a.search_root = t.search_root; a.tree_select = t.tree_select;   // (launch_trace_primary switches to quad_cam)
    a.v0 = t.scene->v0; a.e1 = t.scene->e1; a.e2 = t.scene->e2; a.nrm = t.scene->nrm;
    a.pos = t.pos; a.trirec = t.trirec; a.tricam = t.tricam;
    a.ctrl = (TraceCtrl*)t.ctrl;
    a.long_list = t.long_list; a.long_cap = t.long_cap;
    a.n_records = (t.n_leaves - 1u) * (t.search_root ? 2u : 1u);
    a.n_leaves = t.n_leaves;
    a.packet_spill = t.packet_spill; a.packet_spill_entries = t.packet_spill_entries; a.packet_spill_slots = t.packet_spill_slots; a.diagnostics = t.diagnostics ? 1u : 0u;
    a.long_batch = t.long_batch;            // knob RTMI_LONG_BATCH (0: by the number of deferred rays)
    a.keys = t.keys; a.split_list = t.split_list; a.pairs = t.pairs;
    a.task_shards = (TaskShard*)t.task_shards; a.task_cap = t.task_cap;
    a.defer_budget = t.defer_budget ? t.defer_budget : DEFER_BUDGET_DEFAULT;
}

void launch_trace_primary(hipStream_t s, const TraceLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks,
                          bool compact, rt_contact* out)
{
#if RTMI_TRACE_HANDS_OVER
    if (t.device_arith) { launch_trace_primary_devarith(s, t, cam, rank, nranks, compact, out); return; }
#endif
    TraceArgs a{};
    fill_common(t, a);
    a.cam = cam;
    a.rank = rank; a.nranks = nranks;
    a.tiles_x = rt_tiles_x(cam.resX);
    a.tile_xy = t.tile_xy;
    uint32_t total_tiles = a.tiles_x * rt_tiles_y(cam.resY);
    a.compact = compact ? 1u : 0u;
    a.total_work = rt_owned_tiles(total_tiles, rank, nranks) * RT_TILE * RT_TILE;
    a.rays = nullptr;
    a.out = out;
    if (a.quad) a.quad = t.quad_cam;                  // boxes minus the camera position (k_prepare_camera)
    a.quad_octant_records = (a.quad && t.quad_cam_variants == 9u) ? (t.n_leaves - 1u) * (t.search_root ? 2u : 1u) : 0u;
    a.octant_mask = t.octant_mask; a.plane_bits_known = t.plane_bits_known; a.cal_deadline = t.tag == 1 ? t.cal_deadline : 0ull;
    a.block_order = t.block_order; a.block_cost = t.block_cost;
    if (t.prepare_parts) {
        const float origin[3] = { cam.viewTransform[3], cam.viewTransform[7], cam.viewTransform[11] };
        // an eager frame (nothing captured, nothing updated): the control block and the task queues the trace launch starts from are zeroed by
        // the preparation kernel on its way instead of by two fills in front of the trace kernel (launch_tagged)
        const bool eager = !t.graph_record && !t.graph_update;
        const bool four_wide = t.packet && t.packet4 && t.quad && t.tricam && t.search_root;
        const bool shares = four_wide && t.task_shards && t.task_cap && a.defer_budget < (1u << 20);
        const bool wants_ctrl = !(four_wide && !shares && t.no_unsafe_rays);
        if (eager && launch_prepare_camera(s, t, origin, wants_ctrl ? t.ctrl : nullptr, wants_ctrl ? trace_ctrl_error_offset() : 0u,
                                           shares ? t.task_shards : nullptr, shares ? trace_task_bytes(t.task_cap) : 0u))
            a.cleared_by_preparation = 1u;
        else if (!eager) launch_prepare_camera(s, t, origin);
    }
    launch<1>(s, t, a);
}

// the per-camera records: t.prepare_parts says which (see k_prepare_camera)
bool launch_prepare_camera(hipStream_t s, const TraceLaunch& t, const float origin[3], void* clear_a, size_t clear_a_bytes, void* clear_b, size_t clear_b_bytes)
{
#if RTMI_TRACE_HANDS_OVER
    if (t.device_arith) return launch_prepare_camera_devarith(s, t, origin, clear_a, clear_a_bytes, clear_b, clear_b_bytes);
#endif
    if (t.n_leaves < 2 || t.prepare_parts == 0u) return false;
    PrepareArgs a{};
    a.clear_a = (uint32_t*)clear_a; a.clear_a_words = (uint32_t)(clear_a_bytes / 4u); a.clear_b = (uint32_t*)clear_b; a.clear_b_words = (uint32_t)(clear_b_bytes / 4u);
    const uint32_t n_ref = t.n_leaves - 1, n_all = n_ref * (t.search_root ? 2u : 1u);
    a.trav = const_cast<TravNode*>(t.trav); a.n_records = n_all; a.n_reference = n_ref;
    a.quad = t.quad; a.quad_cam = const_cast<QuadNode*>(t.quad_cam);
    // the four-wide records are walked on the search tree only (the reference tree's are never read: see k_trace's four-wide step)
    a.quad_first = t.search_root ? t.search_root : 0u; a.quad_count = n_all - a.quad_first; a.copy_stride = n_all;
    a.trirec = t.trirec; a.tricam = const_cast<TriCam*>(t.tricam); a.n_tris = t.n_leaves;
    a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2];
    a.select = const_cast<uint32_t*>(t.tree_select);
    a.parts = t.prepare_parts;
    if (!t.quad || !t.quad_cam) a.parts &= ~(RT_PREP_GENERAL | RT_PREP_OCTANTS(0xFFu));
    if (t.quad_cam_variants != 9u) a.parts &= ~RT_PREP_OCTANTS(0xFFu);
    if (!t.tricam || !t.trirec) a.parts &= ~RT_PREP_TRIANGLES;
    a.blocks_quads = (a.parts & (RT_PREP_GENERAL | RT_PREP_OCTANTS(0xFFu))) ? (uint32_t)(((uint64_t)a.quad_count * 8u + 255u) / 256u) : 0u;
    a.blocks_marks = (a.parts & RT_PREP_MARKS) ? (n_all + 255u) / 256u : 0u;
    const uint32_t blocks_tris = (a.parts & RT_PREP_TRIANGLES) ? (t.n_leaves + 255u) / 256u : 0u;
    const uint32_t grid = a.blocks_quads + a.blocks_marks + blocks_tris;
    if (grid == 0u) return false;
    if ((a.parts & RT_PREP_MARKS) && !t.graph_update) (void)hipMemsetAsync(a.select, 0, sizeof(uint32_t), s);
    rt_launch_node(t.graph_record, t.graph_update, RT_NODE_PREP_CAMERA, k_prepare_camera, dim3(grid), dim3(256), s, a);
    return true;
}

// Start order of the 8x8 blocks: a launch of one wave per block ends with its longest blocks, which the dispatcher hands out
// wherever they stand in index order (wave timelines: ManFigure's heaviest blocks start 44-75 us into a 277 us launch and end it).
// Frames follow each other, so the previous frame knows: the four-wide kernel notes every block's steps, and every few frames
// the blocks are sorted by that cost, longest first (the library's bitonic network on (~cost, block) pairs); wave w then walks
// block order[w].  Which block a wave walks changes nothing in any contact.
#define BLOCK_ORDER_GROUPS 4096u          // groups one workgroup sorts in LDS (32 KB of pairs)
// One workgroup, one launch: the groups' costs -> keys -> bitonic network in LDS -> order[].  `1 << group_log` consecutive blocks
// keep their places next to each other: neighbouring blocks walk the same records, and consecutive waves land on the same CU
// (scalar cache) and XCD (L2).  Keys ascending = costs descending, ties by index (the pair is compared as a whole, so the network's
// result does not depend on its shape); p2 = groups rounded up to a power of two, the padding sorts behind every group.
__global__ __launch_bounds__(1024) void k_block_order(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t blocks,
                                                      uint32_t group_log, uint32_t groups, uint32_t p2, uint32_t identity)
{
    __shared__ uint2 t[BLOCK_ORDER_GROUPS];
    if (identity) { for (uint32_t i = threadIdx.x; i < blocks; i += 1024u) order[i] = i; return; }
    for (uint32_t i = threadIdx.x; i < p2; i += 1024u) {
        uint32_t c = 0;
        if (i < groups) for (uint32_t k = 0; k < (1u << group_log); k++) { const uint32_t v = cost[(i << group_log) + k]; c = c + v < c ? 0xFFFFFFFEu : c + v; }
        t[i] = i < groups ? make_uint2(0xFFFFFFFEu - (c < 0xFFFFFFFEu ? c : 0xFFFFFFFEu), i) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    __syncthreads();
    for (uint32_t len = 2; len <= p2; len <<= 1)
        for (uint32_t inc = len >> 1; inc; inc >>= 1) {
            for (uint32_t i = threadIdx.x; i < (p2 >> 1); i += 1024u) {
                const uint32_t lo = (i << 1) - (i & (inc - 1u)), hi = lo + inc;
                const uint2 a = t[lo], b = t[hi];
                const bool greater = a.x > b.x || (a.x == b.x && a.y > b.y);
                if (greater == !(lo & len)) { t[lo] = b; t[hi] = a; }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < blocks; i += 1024u) order[i] = (t[i >> group_log].y << group_log) | (i & ((1u << group_log) - 1u));
}
// group_log: as asked for, raised until the frame's groups fit one workgroup; false = a frame too large for that (blocks stay in index order)
bool launch_block_order(hipStream_t s, const uint32_t* cost, uint32_t* order, uint32_t blocks, uint32_t group_log, bool identity)
{
    if (!blocks) return false;
    while (group_log < 6u && (blocks >> group_log) > BLOCK_ORDER_GROUPS) group_log++;
    const uint32_t groups = blocks >> group_log;             // the tiled enumeration's blocks come in multiples of 64 (one 64x64 tile)
    if (groups > BLOCK_ORDER_GROUPS || (groups << group_log) != blocks) return false;
    uint32_t p2 = 1; while (p2 < groups) p2 <<= 1;
    hipLaunchKernelGGL(k_block_order, dim3(1), dim3(1024), 0, s, cost, order, blocks, group_log, groups, p2, identity ? 1u : 0u);
    return true;
}

void launch_trace_rays(hipStream_t s, const TraceLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out)
{
#if RTMI_TRACE_HANDS_OVER
    if (t.device_arith) { launch_trace_rays_devarith(s, t, rays, count, out); return; }
#endif
    TraceArgs a{};
    fill_common(t, a);
    a.total_work = count;
    a.rays = rays;
    a.out = out;
    a.nranks = 1;
    // arbitrary rays may use the search tree too, each under its own guard (ray_root)
    launch<0>(s, t, a);
}

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_trace()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_device_clock)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;           // (the RT_MODE_DEVICE_ARITH instances are loaded where a BVH of that mode is created: rt_bvh_create)
}
