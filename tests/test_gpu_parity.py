"""-m gpu: the hipcc/gfx950 build of the hot path on a real MI355X, through the C ABI, against
the reference-generated fixtures and the oracle on the same seeded inputs."""
import pytest
import torch

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def native():
    from style import _native as nat
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return nat.get()       # raises if libmst_amd.so is missing: no fallback


@pytest.mark.parametrize('name', ['small_unpitched', 'small_pitched_only'])
def test_golden_small(native, name):
    pc.golden_small(native, torch.device('cuda:0'), name)


@pytest.mark.parametrize('C,R,T,unp', [(1, 1, 1, True), (3, 2, 3, False), (2, 5, 1, True)])
def test_oracle_small_widths_ragged_shapes(native, C, R, T, unp):
    pc.oracle_case(native, torch.device('cuda:0'), pc.SMALL, C, R, T, unp, density=0.05, check_bitwise=True)


def test_total_loss_normalize_false_and_true_against_reference_fixture(native):
    pc.loss_normalize_case(native, torch.device('cuda:0'))


def test_oracle_full_widths_small_clip(native):
    pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 2, 2, 4, True, check_bitwise=True)


def test_oracle_bench_clip(native):
    # BASELINE.json configs[1]: one "30 s" clip = (C=4, R=16, T=4), full widths
    e, worst = pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 4, 16, 4, True, check_bitwise=True)
    print('bench clip: all-gradient rel-L2', e, 'worst tensor', worst)


def test_oracle_pitched_only_many_channels(native):
    pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 8, 3, 3, False)


def test_oracle_long_clip_config5_shape(native):
    # BASELINE.json configs[4] on ONE GPU: C=8 channels, R=151 bars (5 min at 120 bpm + 1), T=4 — exercises the
    # 32-bit index math, the 256-slab weight-gradient splits and long (151-step) LSTM chains at 9.4x the bench clip
    e, worst = pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 8, 151, 4, True, clip_id=9)
    print('long clip: all-gradient rel-L2', e, 'worst tensor', worst)


@pytest.mark.parametrize('C,R,T,unp,K', [(2, 2, 2, True, 3), (1, 3, 1, False, 2)])
def test_batched_clips_small(native, C, R, T, unp, K):
    pc.batch_case(native, torch.device('cuda:0'), pc.SMALL, C, R, T, unp, K)


@pytest.mark.parametrize('K,tile', [(4, None), (5, None), (5, 64)])
def test_batched_clips_full_widths(native, K, tile):
    # BASELINE.json configs[2] in miniature: K different clips in one plan, every launch carrying all of them; K = 4, 5 run the
    # product's default tiling for few clips per launch (32x32 split-K, batched), (5, 64) the 64x64 tiling
    pc.batch_case(native, torch.device('cuda:0'), pc.FULL, 3, 4, 2, True, K, gemm_tile=tile)


def test_large_linear_kernels_on_every_eligible_layer(native):
    # lin.hip (2 x 2-blocked MFMA tiles, all clips as rows of one launch) takes the large Linears (>= 512 rows, >= 4 MFLOP per clip) of the bench plans (the two
    # tests below); dense_flavour = 2 sends every eligible Linear there: ragged row totals, N from 16 to 376, K from 4 to 514
    pc.batch_case(native, torch.device('cuda:0'), pc.FULL, 3, 4, 2, True, 5, gemm_tile=64, dense_flavour=2)
    pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 7, 3, 3, True, gemm_tile=64, dense_flavour=2)


def test_branches_on_four_streams_match_the_oracle(native):
    # mst_plan_options.branches = 1, eager launches: the passes really run on the caller's stream + three side streams with event
    # waits where a dependency crosses streams — a missing dependency shows as a wrong gradient here (three shapes, repeated)
    for rep in range(3):
        pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 4, 16, 4, True, branches=1)
    pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 7, 3, 3, True, branches=1)
    pc.batch_case(native, torch.device('cuda:0'), pc.FULL, 3, 4, 2, True, 3, branches=1)


def test_batched_32_bench_clips_configs3_per_gpu_share(native):
    # BASELINE.json configs[3]: minibatch 256 over 8 GPUs = 32 x (C=4, R=16, T=4) clips per GPU in one plan; against the
    # oracle clip by clip (outputs, 15 loss leaves per clip, summed gradient) and against 32 one-clip iterations bit for bit
    pc.batch_case(native, torch.device('cuda:0'), pc.FULL, 4, 16, 4, True, 32)


def test_batched_64_bench_clips_configs2(native):
    # BASELINE.json configs[2] at full size: 64 x (C=4, R=16, T=4) clips in one plan == 64 one-clip iterations bit for bit,
    # and == the oracle's 64 accumulated iterations (every clip's outputs and losses, the summed gradient)
    pc.batch_case(native, torch.device('cuda:0'), pc.FULL, 4, 16, 4, True, 64)


def test_single_clip_on_the_mfma_gemm(native, monkeypatch):
    monkeypatch.setenv('MST_GEMM', 'mfma')
    pc.golden_small(native, torch.device('cuda:0'), 'small_unpitched')
    e, worst = pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 4, 16, 4, True, check_bitwise=True)
    print('bench clip on MFMA: all-gradient rel-L2', e, 'worst tensor', worst)


def test_oracle_training_cap_shape(native):
    # the reference's own cap: songs are cut to 800 // C bars for training (train-model.py:101) => C=4, R=200, T=4
    e, worst = pc.oracle_case(native, torch.device('cuda:0'), pc.FULL, 4, 200, 4, True, clip_id=11)
    print('training cap: all-gradient rel-L2', e, 'worst tensor', worst)


def test_bar_tiling_two_tiles_bench_clip(native):
    # the bench clip's 16 bars over two "ranks" (both plans on this GPU, exchanges summed by hand)
    from test_tiled import run_tiled
    from oracle import style_oracle as so
    from tools.synth import synth_clip
    from simutil import make_dims, rel
    from style import _native as nat
    dev = torch.device('cuda:0')
    C, R, T = 4, 16, 4
    dims = make_dims(pc.FULL, C, R, T, True)
    flat, named, table = pc.random_params(native, dims, 0)
    clip = synth_clip(5, C, R, T, True)
    gt, lt, plans, nx = run_tiled(native, dev, pc.FULL, C, R, T, True, [(0, 9), (9, 7)], clip, flat)
    _, ref_losses = so.iteration(named, clip)
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1) for n, _, _ in table])
    assert rel(gt.cpu().numpy(), gref.numpy()) < pc.TOL
    for l in lt:
        for i, k in enumerate(nat.LOSS_KEYS):
            if k in ref_losses:
                assert abs(float(l[i]) - ref_losses[k]) < 5e-5, k


def test_bar_tiling_config5_shape_eight_tiles(native):
    # BASELINE.json configs[4]: C=8, R=151 bars (5 min + 1), T=4, bars tiled 8 ways (19 bars per rank, 18 for the last);
    # eight plans on this one GPU stand in for the eight ranks
    from test_tiled import run_tiled
    from oracle import style_oracle as so
    from tools.synth import synth_clip
    from simutil import make_dims, rel
    dev = torch.device('cuda:0')
    C, R, T = 8, 151, 4
    dims = make_dims(pc.FULL, C, R, T, True)
    flat, named, table = pc.random_params(native, dims, 0)
    clip = synth_clip(9, C, R, T, True)
    tiles = [(19 * k, 19 if k < 7 else 18) for k in range(8)]
    gt, lt, plans, nx = run_tiled(native, dev, pc.FULL, C, R, T, True, tiles, clip, flat)
    _, ref_losses = so.iteration(named, clip)
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1) for n, _, _ in table])
    e = rel(gt.cpu().numpy(), gref.numpy())
    print('configs[4] shape, 8 tiles: all-gradient rel-L2', e, 'exchanges per iteration', nx)
    assert e < pc.TOL
    assert abs(float(lt[3][0]) - ref_losses['total']) < 5e-5
