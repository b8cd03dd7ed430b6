"""The grid walk divides by short multiply-add sequences (tlg_kernels.hip: `div_by`, five operations given RN(1 / y); `div_by_res`, three, for
y = (float)res) instead of the compiler's IEEE division.  Their claim -- the same bits as x / y for operands between 2^-90 and 2^90 -- is checked
exhaustively on the GPU by tools/micro/div_check.hip (profiles/r05_division_sequences_check.txt); here the same arithmetic in C with fmaf against this
CPU's division on pseudo-random operands (tests/helpers/div_sequences.c): no GPU, no oracle."""
import ctypes as C
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    out = os.path.join(tempfile.mkdtemp(prefix='rtmi_divseq_'), 'libdivseq.so')
    # -ffp-contract=off: the sequences spell their fused operations out (fmaf), nothing else may fuse; -mfma so that fmaf is the instruction
    subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-mfma', '-shared', '-fPIC', '-o', out, os.path.join(HERE, 'helpers', 'div_sequences.c'), '-lm'])
    lib = C.CDLL(out)
    lib.div_sequences_mismatches.restype = C.c_uint64
    lib.div_sequences_mismatches.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]
    return lib


def test_three_operations_divide_by_a_leaf_resolution():
    lib = _lib()
    first = (C.c_uint32 * 2)()
    for seed, res_max in ((1, 8), (2, 1023), (3, 2048)):
        bad = lib.div_sequences_mismatches(0, res_max, 20_000_000, seed, first)
        assert bad == 0, (res_max, hex(first[0]), hex(first[1]))


def test_five_operations_divide_by_any_plane_spacing():
    lib = _lib()
    first = (C.c_uint32 * 2)()
    bad = lib.div_sequences_mismatches(1, 0, 60_000_000, 7, first)
    assert bad == 0, (hex(first[0]), hex(first[1]))
