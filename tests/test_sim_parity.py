"""The product's HIP kernels executed by the hipsim CPU interpreter (tests/hipsim) through the
product's C ABI, against (a) fixtures produced by the reference and (b) the oracle on seeded
inputs.  CPU-only rehearsal of tests/test_gpu_parity.py: same sources, same ABI, no GPU."""
import pytest

import parity_cases as pc
from simutil import sim_native


@pytest.mark.parametrize('name', ['small_unpitched', 'small_pitched_only'])
def test_golden_small(name):
    pc.golden_small(sim_native(), 'cpu', name)


@pytest.mark.parametrize('C,R,T,unp', [(1, 1, 1, True), (3, 2, 3, False), (2, 5, 1, True), (3, 3, 2, True)])
def test_oracle_small_widths_ragged_shapes(C, R, T, unp):
    # C * R >= 8 sequences per beat LSTM take the grouped kernels (4 sequences per workgroup) on the interpreter: 10 = 4 + 4 + 2, 9 = 4 + 4 + 1
    pc.oracle_case(sim_native(), 'cpu', pc.SMALL, C, R, T, unp, density=0.05, check_bitwise=True)


def test_oracle_full_widths():
    pc.oracle_case(sim_native(), 'cpu', pc.FULL, 2, 2, 2, True, density=0.03)


def test_empty_clip_has_finite_losses():
    # all-zero piano roll (train-model.py:105-106 skips these; the kernels must still not fault):
    # mask.sum() == 0 makes the masked means 0/0 = NaN in the reference too
    import torch
    from style import _native as nat
    native = sim_native()
    dims = pc.make_dims(pc.SMALL, 1, 2, 2, False)
    flat, named, table = pc.random_params(native, dims)
    plan = nat.Plan(native, dims, 'cpu')
    plan.set_inputs(mode=[1., 0.], bpm=[120.], instr=torch.zeros(1, 51), used=torch.zeros(1, 41), bpm_target=120.)
    g = torch.zeros_like(flat)
    losses = torch.zeros(nat.N_LOSSES)
    plan.train_iteration(flat, g, torch.zeros(1, 1, 2, 2, 10, 56, 5), None, losses)
    assert torch.isfinite(plan.view('pitched_pred')).all()


@pytest.mark.parametrize('C,R,T,unp,K', [(2, 2, 2, True, 3), (1, 3, 1, False, 2)])
def test_batched_clips_equal_sequential_iterations(C, R, T, unp, K):
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, C, R, T, unp, K)


def test_batched_clips_on_the_mfma_gemm():
    # the product switches to the 64x64-tile GEMM at 6 clips per launch; force it for a 4-clip plan, and run the
    # default (32x32 tiles with 4 clips per launch) as well
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 2, 2, 1, True, 4, gemm_tile=64)
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 2, 2, 1, True, 4)
    # 9 positions x 8 octaves = 72 conv rows per clip: the conv weight gradient folds the clips too, and its 32-row k-tiles
    # cross clip boundaries (72 is not a multiple of 32)
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 3, 3, 1, True, 3, gemm_tile=64)


def test_mfma_gemm_runs_of_several_tiles_per_workgroup():
    # the 64x64 tiling walks a run of output tiles per workgroup (prefetching across tile boundaries); at test sizes the
    # automatic choice is one tile, so force runs of 3 (tile counts that are not multiples of it included)
    pc.oracle_case(sim_native(), 'cpu', pc.FULL, 2, 3, 2, True, density=0.03, gemm_tile=64, gemm_run=3)
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 2, 2, 1, True, 4, gemm_tile=64, gemm_run=2)


def test_large_linear_kernels_at_small_sizes():
    # lin.hip serves the large nn.Linear layers (>= 512 rows and >= 4 MFLOP per clip) of plans on the 64x64 tiling; dense_flavour = 2 sends every eligible
    # layer there: one clip against the oracle (every tensor, every parameter gradient), then ragged clip counts whose row
    # total is not a multiple of the 64/128/256-row tiles and whose weight-gradient row splits cross clip boundaries
    pc.oracle_case(sim_native(), 'cpu', pc.FULL, 2, 3, 2, True, density=0.03, gemm_tile=64, dense_flavour=2)
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 2, 2, 1, True, 4, gemm_tile=64, dense_flavour=2)
    pc.batch_case(sim_native(), 'cpu', pc.SMALL, 3, 3, 1, True, 3, gemm_tile=64, dense_flavour=2)


def test_branches_keep_the_results():
    # the stream assignment only changes where launches are queued; on the interpreter everything is sequential, so this pins the
    # list order it produces (a valid order by construction) against the oracle and the bitwise golden comparison
    pc.oracle_case(sim_native(), 'cpu', pc.SMALL, 3, 2, 3, True, density=0.05, check_bitwise=True, branches=1)


def test_single_clip_on_the_mfma_gemm(monkeypatch):
    monkeypatch.setenv('MST_GEMM', 'mfma')
    pc.oracle_case(sim_native(), 'cpu', pc.SMALL, 3, 2, 3, True, density=0.05, check_bitwise=True)
    pc.golden_small(sim_native(), 'cpu', 'small_unpitched')


def test_total_loss_normalize_false_and_true_against_reference_fixture():
    pc.loss_normalize_case(sim_native(), 'cpu')
