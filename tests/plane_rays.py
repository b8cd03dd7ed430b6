"""Test helper (numpy, CPU): which leaves does a ray REACH -- (a) in the reference tree under the reference's staged slab test
(AABB.h:126-142 + BVH.h:366-372, NaN behaviour included: comparisons with NaN are false, the ternary min / max macros of
CLPortability.h:147-148 return their right operand on NaN), (b) in any other hierarchy of nested boxes over the same leaves
under the test the four-wide packet kernel evaluates on its per-octant records (products of the near / far planes, max3 / min3
that IGNORE NaN, `near <= far`, `near > threshold`).

Used to check the claim DESIGN.md section 2 makes for rays whose only `0 * inf` sits on the x axis: NaN in the FIRST axis of the
staged test drops out exactly as -inf (near) / +inf (far) would, which makes the test monotone under nesting again, so such
rays may stay on the search tree.  On y or z the staged test loses earlier constraints and the result depends on the hierarchy.
"""
import numpy as np

f32 = np.float32


def staged_valid(lo, hi, o, inv):
    """reference child test for boxes lo/hi [k,3] and one ray (o[3], inv[3] = 1/direction); numpy fp32, NaN as in C"""
    with np.errstate(all='ignore'):
        s = [f32(0.0) > inv[k] for k in range(3)]
        near = [(hi[:, k] if s[k] else lo[:, k]) for k in range(3)]
        far = [(lo[:, k] if s[k] else hi[:, k]) for k in range(3)]
        tmin = (near[0] - o[0]) * inv[0]
        tmax = (far[0] - o[0]) * inv[0]
        tymin = (near[1] - o[1]) * inv[1]
        tymax = (far[1] - o[1]) * inv[1]
        flag = ~((tmin > tymax) | (tymin > tmax))
        tmin = np.where(tmin > tymin, tmin, tymin)            # max(a, b) = a > b ? a : b
        tmax = np.where(tmax < tymax, tmax, tymax)            # min(a, b) = a < b ? a : b
        tzmin = (near[2] - o[2]) * inv[2]
        tzmax = (far[2] - o[2]) * inv[2]
        flag &= ~((tmin > tzmax) | (tzmin > tmax))
        tmin = np.where(tmin > tzmin, tmin, tzmin)
        inside = ((lo[:, 0] <= o[0]) & (o[0] <= hi[:, 0]) & (lo[:, 1] <= o[1]) & (o[1] <= hi[:, 1]) &
                  (lo[:, 2] <= o[2]) & (o[2] <= hi[:, 2]))
        return (flag & (tmin > 0)) | inside


def octant_valid(lo, hi, o, inv):
    """the four-wide packet kernel's test on its per-octant, camera-relative records (trace_kernels.hip, child_mask_octant)"""
    with np.errstate(all='ignore'):
        s = [f32(0.0) > inv[k] for k in range(3)]
        n = [((hi[:, k] if s[k] else lo[:, k]) - o[k]) * inv[k] for k in range(3)]
        f = [((lo[:, k] if s[k] else hi[:, k]) - o[k]) * inv[k] for k in range(3)]
        near = np.fmax(np.fmax(n[0], n[1]), n[2])             # v_max3_f32: a NaN operand is ignored
        far = np.fmin(np.fmin(f[0], f[1]), f[2])
        inside = ((lo[:, 0] <= o[0]) & (o[0] <= hi[:, 0]) & (lo[:, 1] <= o[1]) & (o[1] <= hi[:, 1]) &
                  (lo[:, 2] <= o[2]) & (o[2] <= hi[:, 2]))
        thr = np.where(inside, f32(-np.inf), f32(0.0))
        return (near <= far) & (near > thr)


def reached_leaves(nodes, n, valid):
    """leaves [0, n) that are valid and whose every ancestor is valid; nodes: oracle NODE array, root = n (entered without a
    test, BVH.h:352-356), data[0] = parent"""
    parent = nodes['data'][:, 0].astype(np.int64)
    parent[n] = n
    valid = valid.copy()
    valid[n] = True
    reach = valid.copy()
    for _ in range(len(nodes)):
        new = valid & reach[parent]
        if (new == reach).all():
            break
        reach = new
    return np.nonzero(reach[:n])[0]


def plane_camera(osc, res, axis, value, offset=8):
    """a camera whose coordinate `axis` is `value` AND which has rays with a zero direction component on that axis: generateRay
    adds the translation to the direction (Transform.h:282-295), v_k = d . row_k + T_k, so with row_k = (+-T_k / offset on the
    diagonal, 0, 0) the pixel column / row `offset` away from the centre has v_k = 0 exactly (offset a power of two); for z every
    ray has (FOVDistance = offset).  Anisotropic, but a valid Camera."""
    cam = osc.demo_camera(*res).copy()
    m = cam['m'][0]
    value = f32(value)
    if axis == 0:
        m[0] = value / f32(offset); m[3] = value
    elif axis == 1:
        m[5] = -value / f32(offset); m[7] = value
    else:
        cam['FOVDistance'][0] = f32(offset); m[10] = -value / f32(offset); m[11] = value
    return cam
