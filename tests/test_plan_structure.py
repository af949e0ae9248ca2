"""Structure of the static plans, read through the C ABI's instrumentation entry points (mst_plan_step_count,
mst_plan_step_gemms) on the CPU interpreter build: what the scheduler is documented to do is visible in
the launch lists, so a regression that silently falls back to the unfused shapes fails here and not only in a profile."""
import parity_cases as pc
from simutil import sim_native
from style import _native as nat


def gemm_members(plan, backward):
    out = []
    n = plan.lib.mst_plan_step_count(plan.handle, 7, int(backward))
    for st in range(n):
        out.extend(plan.step_gemms(7, backward, st))
    return out


def test_batched_plan_folds_the_clips_into_its_weight_gradient_gemms():
    native = sim_native()
    dims = pc.make_dims(pc.FULL, 2, 3, 2, True)
    dims.clips = 4
    plan = nat.Plan(native, dims, 'cpu', gemm_tile=64)
    assert plan.gemm_tile == 64
    bwd = gemm_members(plan, True)
    folded = [m for m in bwd if m[4] > 0]
    assert folded, 'no folded weight-gradient GEMM in a 4-clip plan on the 64x64 tiling'
    for M, N, K, splits, fold_rows, wgs in folded:
        assert K == fold_rows * 4                    # reduction index = (clip, row)
        assert 1 <= splits <= 512 and wgs >= splits  # at least one workgroup per k-split
    # the one-clip plan of the same shape folds nothing
    dims1 = pc.make_dims(pc.FULL, 2, 3, 2, True)
    one = nat.Plan(native, dims1, 'cpu')
    assert all(m[4] == 0 for m in gemm_members(one, True))


def test_batched_plan_runs_one_row_per_clip_linears_with_the_clips_as_rows():
    # the style / song-info heads see ONE row per clip: on the 64x64 tiling an 11-clip plan carries them as GEMMs with M = 11
    # (GemmDesc.clip_rows: one descriptor for all clips, weights read once) in forward and backward; the one-clip plan of the
    # same shape has M = 1 there and no M = 11 anywhere (no layer width, channel, bar or beat count of this shape is 11)
    native = sim_native()
    dims = pc.make_dims(pc.FULL, 3, 2, 2, True)
    dims.clips = 11
    plan = nat.Plan(native, dims, 'cpu', gemm_tile=64)
    for backward in (False, True):
        rows11 = [m for m in gemm_members(plan, backward) if m[0] == 11 and m[4] == 0]
        assert rows11, 'no clips-as-rows GEMM in the %s pass' % ('backward' if backward else 'forward')
    one = nat.Plan(native, pc.make_dims(pc.FULL, 3, 2, 2, True), 'cpu', gemm_tile=64)
    assert not [m for m in gemm_members(one, False) + gemm_members(one, True) if m[0] == 11]


def step_rows(plan, backward, mask=7):
    import numpy as np
    n = plan.lib.mst_plan_step_count(plan.handle, mask, int(backward))
    info = np.zeros((n, 8), np.int32)
    assert plan.lib.mst_plan_step_info(plan.handle, mask, int(backward), info.ctypes.data) == n
    return info.tolist()


def test_batched_plan_runs_its_large_linears_on_the_blocked_tiles():
    # 64 clips, 7 instruments: every large Linear (>= 512 rows, >= 4 MFLOP per clip, more than 32 outputs: the applier's 800-row
    # layers, the 560-row K = 514 / 112 ones) and the rhythm encoder's 280 -> 16 over 5600 note rows is a lin.hip step (kinds 29-31) with all clips as rows of one launch; a one-clip
    # plan on the default (32x32) tiling has none, and dense_flavour = 1 opts out
    native = sim_native()
    dims = pc.make_dims(pc.FULL, 7, 8, 10, True)
    dims.clips = 64
    plan = nat.Plan(native, dims, 'cpu')
    fwd = [r for r in step_rows(plan, False) if r[5] == 29]
    bwd = [r for r in step_rows(plan, True) if r[5] in (30, 31)]
    assert fwd and bwd
    for rows, N, K, splits, count, kind, level, chain in fwd + bwd:
        # the blocked tiles' rule, or the <= 16-output stream kernels' (280 -> 16 over every note row)
        assert rows >= 512 and ((N > 32 and 2.0 * rows * N * K >= 4e6) or (N <= 16 and 128 <= K < 320))
    assert len([r for r in bwd if r[5] == 31]) == len(fwd)               # one weight-gradient launch per forward launch
    off = nat.Plan(native, dims, 'cpu', dense_flavour=1)
    assert not [r for r in step_rows(off, False) + step_rows(off, True) if r[5] in (29, 30, 31)]
    one = nat.Plan(native, pc.make_dims(pc.FULL, 7, 8, 10, True), 'cpu')
    assert not [r for r in step_rows(one, False) + step_rows(one, True) if r[5] in (29, 30, 31)]


def test_rhythm_encoder_linear_never_sees_the_135_wide_concat():
    # pitched rhythm encoder (full widths): cat_with_broadcast of six segments, 135 columns in all, feeding a Linear to 32.
    # Decomposed (linear_bcast), the only full-row GEMM of that Linear has K = 16 (the per-fraction channels block);
    # the materialised version had a (rows x 32 x 135) forward GEMM
    native = sim_native()
    C, R, T = 2, 3, 2
    plan = nat.Plan(native, pc.make_dims(pc.FULL, C, R, T, True), 'cpu')
    fwd = gemm_members(plan, False)
    rows = C * R * T * 10
    assert not [m for m in fwd if m[2] == 135], 'a forward GEMM reduces over the 135-wide concat'
    assert [m for m in fwd if m[0] == rows and m[1] == 32 and m[2] == 16], 'the channels block GEMM (rows x 32 x 16) is missing'


def test_single_launch_slab_reduce_guard_counts_columns_from_the_matrix_base():
    # two column blocks of one (rows x 128) parameter matrix, offsets from the matrix's own element (0, 0) — the plan passes
    # `dst - base`, so a base that is not a multiple of the pitch no longer shifts the columns (base % ld = 100,
    # cols [0, 40) vs [30, 50) were judged disjoint when the ABSOLUTE offset was taken modulo the pitch)
    f = sim_native().lib.mst_debug_slab_columns_disjoint
    assert f(0, 40, 30, 20, 128) == 0            # [0, 40) and [30, 50) overlap
    assert f(0, 30, 30, 20, 128) == 1            # [0, 30) and [30, 50) do not
    assert f(0, 40, 128 + 30, 20, 128) == 0      # blocks starting in different rows: still the same columns
    assert f(100, 40, 10, 20, 128) == 0          # a block that wraps a row is not a column block: never judged disjoint
    assert f(88, 40, 0, 88, 128) == 1


def test_branches_spread_the_whole_model_lists_over_streams():
    # mst_plan_options.branches = 1: every launch of the whole-model lists carries a stream (0 = the caller's, 1..3 = side
    # streams) chosen along the dependency DAG; plan creation itself checks that every pair of launches that share memory
    # is ordered by stream order and event waits (it falls back to one stream otherwise, which this test would see).
    # Default plans keep one stream (chain -1 everywhere); per-stage lists are never spread.
    native = sim_native()
    dims = pc.make_dims(pc.FULL, 4, 16, 4, True)
    spread = nat.Plan(native, dims, 'cpu', branches=1)
    for backward in (False, True):
        streams = [r[7] for r in step_rows(spread, backward)]
        assert set(streams) <= {0, 1, 2, 3} and len(set(streams)) >= 2, streams
        assert all(r[7] == -1 for r in step_rows(spread, backward, mask=1))
    plain = nat.Plan(native, dims, 'cpu')
    assert all(r[7] == -1 for r in step_rows(plain, False) + step_rows(plain, True))
