"""CPU: the HIP kernel files' OWN SOURCE executed on the host.  tools/hostsim.py compiles every file of csrc/ as plain C++ against a
stand-in for the pieces of the HIP programming model they use (one host thread per lane, pthread barriers for workgroup and wave;
shuffles, DPP, ballots and the MFMA instructions as collective operations of a wave; LDS-DMA as a wave-wide copy; the GEMM files' few
inline-assembly helpers get a C++ body in the compiled copy) into a library with the same C ABI -- every entry point -- and the kernel
cases of tests/test_gpu_kernels.py -- the very functions that run on the MI355X -- are run against it here: encoder / unfold, depthwise forward / backward, gLN statistics / apply / backward pieces, head backward,
decoder forward / backward, channel softmax, cLN, SI-SDR, PIT search, Sinkhorn, row distances, squared norm + Adam, chunking /
overlap-add, the LSTM sweeps (both kernels: sixteen and four sequences per workgroup, forced per call),
and the GEMM family in its three arithmetics and the packed-weight form (per-wave split kernels, the cooperative and the producer /
consumer kernels with their flag synchronisation, the weight-gradient kernels, prologues / epilogues, ragged tiles).
It catches indexing / synchronisation / unwritten-output mistakes in the kernel source before any GPU minute is spent.  It cannot see
timing, asynchrony (DMA lands at once, waits are no-ops) or the hardware's operand layouts themselves (tools/mfma4x4_probe.hip)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hostsim                         # noqa: E402
import test_gpu_kernels as GK          # noqa: E402

pytestmark = pytest.mark.skipif(hostsim.compiler() is None, reason="needs clang++ (ext_vector_type)")

CASES = [
    ("test_encoder_and_unfold", [(1, 16, 8, 0, 0), (1, 16, 8, 1, 3), (2, 20, 10, 1, 4), (1, 2, 1, 0, 0)]),
    ("test_dwconv_fwd_bwd", [(300, 8), (1030, 64), (2100, 256), (700, 2), (5000, 1)]),
    ("test_depthwise_generic", [(5, 2, 4, 2, 130), (4, 4, 0, 1, 64), (3, 1, 8, 8, 1000)]),
    ("test_depthwise_tcn_geometry", [(3, 1, 384, True), (3, 4, 512, True), (3, 64, 1024, False), (5, 2, 384, True)]),
    ("test_gln_bwd_finalize", [(2, 8), (8, 1)]),
    ("test_gln_bwd_finalize_batch", [()]),
    ("test_head_bwd", [(0,), (1,)]),
    ("test_decoder_fwd_bwd", [(2, 1, 16, 8, 0, True), (3, 1, 16, 8, 3, False), (5, 1, 16, 8, 0, False), (4, 1, 16, 8, 5, True), (2, 2, 20, 10, 4, True),
                              (2, 1, 2, 1, 0, False), (1, 1, 64, 16, 0, False)]),
    ("test_softmax_over_channels", [(2, 128, 300), (3, 50, 64), (2, 7, 1)]),
    ("test_gln_standalone_and_repack", [()]),
    ("test_cln_fwd_bwd", [(2, 24, 203), (3, 128, 3999), (2, 300, 150)]),          # the last: 16-wave tiles of the chained backward
    ("test_prelu_cln_fwd_bwd", [(2, 24, 203, 0.25), (1, 48, 1030, 0.0)]),
    ("test_attention_core_fwd_bwd", [(2, 70, 2, 16, 0.0), (1, 37, 2, 8, 0.0), (1, 130, 1, 32, 0.2), (1, 257, 1, 16, 0.0), (2, 33, 2, 32, 0.1), (1, 64, 1, 16, 0.0), (2, 31, 4, 32, 0.1), (1, 20, 8, 8, 0.0)]),
    ("test_gln_tokens_fwd_bwd", [(3, 250, 64), (5, 37, 16), (1, 7, 1024), (2, 1500, 64)]),          # the last: sliced sequences
    ("test_rownorm_fwd_bwd", [(100, 256, True, 0.0), (33, 256, True, 0.1), (51, 64, False, 0.0), (9, 1024, True, 0.5), (13, 300, True, 0.25), (5, 4, True, 0.0)]),
    ("test_relu_drop_fwd_bwd", [(4096, 0.0), (1028, 0.1), (4, 0.5)]),
    ("test_sisdr_kernels", [(1, 0), (2, 1), (3, 0)]),
    ("test_pit_search", [(2, 0, 1), (3, 1, 1), (4, 0, 0)]),
    ("test_sinkhorn", [(3, 10, 1.0), (5, 200, 1.0), (10, 5, 0.5)]),
    ("test_rowdiff_sums_and_bwd", [(8, 32000), (1, 5)]),
    ("test_sqnorm_and_adam", [()]),
    ("test_segment_overlap_add", [(812, 20, 10), (3999, 250, 125), (100, 16, 4), (64, 64, 64)]),
    ("test_lstm_sweeps", [(16, 5, 7, 0, "sixteen"), (32, 37, 23, 1, "sixteen"), (64, 18, 12, 0, "sixteen"), (128, 21, 9, 1, "sixteen"),       # (lanes are host threads and every step
                          (16, 5, 7, 0, "four"), (32, 37, 23, 1, "four"), (64, 18, 12, 0, "four"), (128, 21, 9, 1, "four")]),                   #  is several barriers: short sequences here,
    ("test_lstm_sweeps_both_directions_one_launch", [("sixteen", 19, 8), ("four", 19, 8)]),                                                   #  the long ones on the device)
    ("test_lstm_sweeps_interleaved_output", [("sixteen",), ("four",)]),
    ("test_chunk_tokens_layout_pair", [(2, 64, 5, 250), (1, 48, 3, 33), (3, 7, 2, 1)]),
    ("test_linear_forward_and_input_gradient", [(1000, 64, 512), (777, 256, 64), (130, 128, 128)]),
    ("test_tcn_layer_kernel_by_kernel_against_the_oracle", [(1, 128, 128, 128, 300, 2), (2, 128, 128, 128, 300, 4)]),
    ("test_head_and_tail_kernel_by_kernel_against_the_oracle", [(1, 128, 128, 128, 2, 1203, True), (2, 128, 128, 128, 3, 1205, False)]),
    ("test_criterion_kernels_against_the_oracle", [(2, 4, 8000), (4, 3, 4001)]),
    ("test_absmax_and_memset", [(1,), (1000,), (70001,)]),
    ("test_linear_weight_gradient", [(32, 20, 64, 512, 0, 7), (9, 31, 128, 512, -1, 3), (9, 31, 128, 512, 1, 5), (3, 7, 64, 64, -1, 1), (5, 250, 256, 64, 0, 40)]),
]


# (arithmetic, test function, arguments without the trailing `arith`); "f16x3-packed" hands over weights split by sep_pack_weights
GEMM_CASES = [
    ("f32", "test_gemm_plain_bias", (2, 64, 128, 300)),
    ("f16x3", "test_gemm_plain_bias", (3, 128, 64, 129)),
    ("f16x3-packed", "test_gemm_plain_bias", (2, 64, 128, 300)),
    ("bf16x6", "test_gemm_plain_bias", (3, 128, 64, 129)),
    ("f16x3-packed", "test_gemm_gln_prologue_and_stats_epilogue", ()),
    ("f16x3-packed", "test_gemm_dgrad_two_sources_rowsums", ()),
    ("f16x3-packed", "test_gemm_dgrad_prelu_bwd", ()),
    ("f16x3-packed", "test_gemm_dgrad_two_sources_plain", (512, 700)),
    ("f16x3-packed", "test_gemm_packed_heads_residual_accumulate", ()),
    ("f16x3", "test_gemm_heads_plain_input_residual_accumulate", ()),           # the staged causal layers' heads and heads^T: the direct kernel, fp32 weights
    ("f16x3", "test_gemm_dgrad_two_sources_plain", (256, 130)),
    ("f32", "test_gemm_gln_bwd_prologue", (0,)),
    ("bf16x6", "test_wgrad_two_sources_gln_prelu", ()),
    ("f16x3", "test_wgrad_latent_product_and_prelu", ()),
    ("f16x3", "test_gemm_small_widths_of_the_dual_path_separators", (48, 48, 77)),
    ("f16x3-packed", "test_gemm_prelu_prologue_sigmoid", ()),
    ("f16x3-packed", "test_gemm_gln_bwd_prologue", (1,)),
    ("f16x3-packed", "test_gemm_gln_bwd_prologue_several_row_tiles", (256, 64)),
    ("f16x3", "test_gemm_gln_bwd_prologue_several_row_tiles", (192, 48)),
    ("f16x3", "test_wgrad_plain", (2, 256, 128, 300, 3)),
    ("bf16x6", "test_wgrad_sample_aligned_slabs_and_gln_sums_from_them", (2, 256, 128, 300, 3)),      # producer / consumer kernel
    ("f32", "test_wgrad_sample_aligned_slabs_and_gln_sums_from_them", (3, 32, 20, 500, 2)),
    ("bf16x6", "test_wgrad_plain", (2, 128, 256, 130, 1)),
    ("f32", "test_wgrad_plain", (2, 32, 4, 201, 2)),
    (None, "test_gemm_packed_weights_model_shapes", (128, 512, 130)),        # producer / consumer kernel, 256-column workgroup tile
    (None, "test_gemm_packed_weights_model_shapes", (512, 128, 130)),        # cooperative kernel
    (None, "test_gemm_packed_weights_model_shapes", (1024, 128, 130)),       # producer / consumer kernel, 4 x 1 consumer waves
    (None, "test_gemm_packed_adversarial_operands", ("late_jump_k1024",)),
    (None, "test_gemm_prelu_prologues_any_slope", (-0.3,)),
    (None, "test_wgrad_f16_adversarial_operands", ("late_jumps",)),
    (None, "test_wgrad_f16_adversarial_operands", ("zero_rows_then_signal",)),
    ("f16x3", "test_wgrad_plain", (2, 512, 128, 999, 7)),                      # the scaled two-part fp16 weight-gradient kernel
    ("f16x3", "test_wgrad_two_sources_gln_prelu", ()),
    (None, "test_wgrad_with_a_presplit_second_source", (2, 128, 300, 2)),            # sep_split_rows + the G2_pre form of the fp16 weight-gradient kernel
    ("f16x3", "test_wgrad_batch_equals_separate_calls", (1, 256, 128, 300, 2, 3)),   # the batched grid of the fp16 producer / consumer kernel
    ("f32", "test_wgrad_batch_equals_separate_calls", (2, 64, 32, 201, 2, 2)),       # ... and the entry point's n-calls fallback
    (None, "test_pack_weights_reproduces_the_weights", ()),
    (None, "test_reduce_slabs_and_f64", ()),
]


@pytest.fixture(scope="module")
def sim_library(tmp_path_factory):
    return hostsim.build(str(tmp_path_factory.mktemp("hostsim")))


@pytest.fixture()
def on_host(sim_library):
    """the kernel tests' device hooks pointed at the host simulation"""
    saved = (GK.HIP, GK.to_device, GK.device_sync, GK.device_name)
    with hostsim.HostSimBackend(sim_library) as K:
        GK.HIP, GK.to_device, GK.device_sync, GK.device_name = K, (lambda t: t.clone()), (lambda: None), (lambda: "cpu")
        try:
            yield K
        finally:
            GK.HIP, GK.to_device, GK.device_sync, GK.device_name = saved


@pytest.mark.parametrize("name,params", CASES, ids=[c[0][5:] for c in CASES])
def test_kernel_source_on_the_host_matches_the_restatement(on_host, name, params):
    for p in params:
        getattr(GK, name)(*p)


@pytest.mark.parametrize("arith,name,args", GEMM_CASES, ids=["{}-{}-{}".format(a or "", n[5:30], "x".join(map(str, g))) for a, n, g in GEMM_CASES])
def test_gemm_kernel_source_on_the_host_matches_the_restatement(on_host, arith, name, args):
    import sepkernels
    if arith is None:
        return getattr(GK, name)(*args)
    prev = sepkernels.set_gemm_arith(arith.split("-")[0])
    GK.PACKED[0] = arith.endswith("packed")
    try:
        getattr(GK, name)(*args, arith)
    finally:
        GK.PACKED[0] = False
        sepkernels.set_gemm_arith(prev)


@pytest.mark.parametrize("config", ["tiny", "softmax"])
def test_whole_fused_conv_tasnet_through_the_kernel_sources(on_host, golden_dir, config):
    """End to end: the fused Conv-TasNet of the product (models/conv_tasnet.py -> sepkernels/net.py orchestration -> C ABI) with the
    host simulation of the kernel sources behind the ABI, on the reference's golden vectors (BASELINE.json configs[0] family: tiny, ReLU
    encoder, 2 speakers; and the same with the channel-softmax mask): forward, PIT loss, permutation and every parameter gradient.
    (`mid` -- 3 speakers, two blocks, skip width != bottleneck width -- passes the same way in 55 s and is left out for time.)  ~30 kernel launches forward, ~60 backward --
    encoder, packed-weight GEMMs with gLN / PReLU prologues and statistics / residual epilogues, depthwise forward / backward, gLN
    finalisation, decoder, weight gradients, slab reduction, SI-SDR pair matrix, permutation search -- none of them emulated."""
    import numpy as np
    import sepkernels
    from oracle.make_golden import CONFIGS
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    from sepkernels import net

    class Named:                                   # the binding object under a name the modules do not take for the GPU build
        name = "hostsim"

        def __getattr__(self, attr):
            return getattr(on_host, attr)
    g = np.load(os.path.join(golden_dir, "convtasnet_{}.npz".format(config)))
    model = ConvTasNet(**CONFIGS[config])
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")})
    assert model.fused
    old = sepkernels._set_backend_for_tests(Named())
    try:
        est = model(torch.from_numpy(g["mixture"]))
        ref = torch.from_numpy(g["output_f64"])
        assert (est.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
        loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, torch.from_numpy(g["sources"]))
        assert abs(loss.item() - float(g["loss_f64"])) <= 1e-5 * abs(float(g["loss_f64"]))
        assert np.array_equal(pattern.numpy(), g["pattern"])
        loss.backward()
    finally:
        sepkernels._set_backend_for_tests(old)
    worst = scale = 0.0
    for k, p in model.named_parameters():
        r = torch.from_numpy(g["grad/" + k]).double()
        worst, scale = max(worst, (p.grad.double() - r).abs().max().item()), max(scale, r.abs().max().item())
    assert worst <= 1e-4 * scale, (worst, scale)


@pytest.mark.parametrize("kind", ["pit", "sinkpit"])
def test_recorded_train_step_replays_through_sep_run_sequence(on_host, kind):
    """ABI 23 end to end on the kernel sources: FusedTrainStep.record runs one training step of the tiny Conv-TasNet (forward, PIT over the
    SI-SDR pair matrix, backward, clip, Adam) while the binding records every launch; the next two steps are ONE sep_run_sequence call
    each (the C loop of csrc/sequence.hip over the recorded ops, a learning-rate change in between through device memory).  Three eager
    steps from the same start give the same losses to the last bit and the same parameters to an ulp: same entry points, same
    arguments, same order -- and sep_memset / sep_absmax / sep_pit_finish do what the eager step's torch kernels do."""
    import sepkernels
    from oracle.make_golden import CONFIGS
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d, SinkPIT
    from sepkernels.train import FusedTrainStep

    class Named:
        name = "hostsim"

        def __getattr__(self, attr):
            return getattr(on_host, attr)
    make = (lambda: PIT1d(NegSISDR(), n_sources=2)) if kind == "pit" else (lambda: SinkPIT(NegSISDR(), n_sources=2, coldness=1.0, iteration=7))
    g = torch.Generator().manual_seed(5)
    nsteps = 3 if kind == "pit" else 2
    batches = [0.1 * torch.randn(2, 2, 1203 if kind == "pit" else 803, generator=g) for _ in range(nsteps)]
    old = sepkernels._set_backend_for_tests(Named())
    runs = []
    try:
        for recorded in (False, True):
            torch.manual_seed(1)
            model = ConvTasNet(**CONFIGS["tiny"])
            step = FusedTrainStep(model, make(), lr=1e-3, max_norm=5.0)
            losses = []
            for i, src in enumerate(batches):
                mix = src.sum(1, keepdim=True).contiguous()
                if i == nsteps - 1:
                    step.lr = 5e-4
                if recorded and i == 0:
                    losses.append(float(step.record(mix, src)))
                    names = step._seq.names()
                    assert names[0] == "sep_absmax" and names[-1] == "sep_adam_step_dev" and names.count("sep_pit_finish") == 1
                    assert ("sep_sinkhorn_bwd" in names) == (kind == "sinkpit") and ("sep_pit_search" in names) == (kind == "pit")
                    assert "sep_pw_gemm" in names and "sep_pw_wgrad" in names and "sep_memset" in names
                else:
                    losses.append(float(step(mix, src)))
                    assert (step._seq is not None) == recorded
            assert step.step_count == nsteps and (not recorded or int(step._step_dev.item()) == nsteps)
            runs.append((losses, model.flat_parameters().detach().clone(), step.last_pattern if recorded else None))
    finally:
        sepkernels._set_backend_for_tests(old)
    (l0, p0, _), (l1, p1, pattern) = runs
    assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(l0, l1)) and l0[0] != l0[-1] and (kind != "pit" or l0 == l1), (l0, l1)
    assert (p0 - p1).abs().max().item() <= (2e-7 if kind == "pit" else 2e-6) * p0.abs().max().item()
    assert pattern.shape == (2, 2) and sorted(pattern[0].tolist()) == [0, 1]


def test_random_small_shapes_through_the_kernel_sources(on_host):
    """tools/hostsim_fuzz.py with a fixed seed: ragged frame counts and widths off every tile size through GEMMs (four arithmetics incl.
    packed weights, transposed / bias / PReLU prologue / sigmoid epilogue), weight gradients (slabs and one-slab accumulation), encoder /
    decoder forward / backward at random hop geometries, gLN / cLN, chunking, LSTM sweeps.  (Thousands of such cases were run while this
    was written -- 0 failures; the tool takes a duration and a seed.)"""
    import hostsim_fuzz
    n, failures = hostsim_fuzz.run_cases(seed=7, max_cases=120)
    assert not failures, failures[:3]
    assert n == 120


def test_the_comparison_is_not_vacuous(on_host):
    """the same harness fails when the device side computes something else (here: a gain 1 % off)"""
    class Skewed:
        def __getattr__(self, name):
            return getattr(on_host, name)

        def gln_apply(self, x, stats, gamma, *rest):
            return on_host.gln_apply(x, stats, gamma * 1.01, *rest)
    B, C, T, ldt = 1, 4, 130, 256
    x = GK.padded(B, C, T, ldt)
    args = [x, GK.stats_of(x, T), GK.rnd(C) + 1, GK.rnd(C), GK.nan(B, C, ldt), B, C, T, ldt, C * float(T), 1e-12]
    GK.both("gln_apply", list(args))
    GK.HIP = Skewed()
    with pytest.raises(AssertionError):
        GK.both("gln_apply", list(args))
