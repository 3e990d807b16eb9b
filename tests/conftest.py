import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hi3d-official_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# Which GPU tests are the evidence for which row of SURVEY.md 8 (the grading contract).  `pytest -m gpu -x` stops at the first
# failure, so the ORDER decides what a single broken test can hide: round 4 ended with an opt-in experiment's kernel test
# (alphabetically early) in front of every VideoDecoder / checkpoint / torch.ops / clip-parallel test, and 150 tests never ran.
# pytest_collection_modifyitems below runs (0) the row evidence in this order, (1) the remaining kernel / property tests,
# (2) stress screens, 1000-launch repeatability runs and tests of opt-in (default-off) code paths -- last.
# A node id matches an entry when the entry is a substring of it.  tests/test_host_cpu.py checks that every row has a test.
ROW_TESTS = [
    ("a1-a6 sampler step / guider / denoiser / wrapper", [
        "test_unet_gpu.py::test_sampler_matches_reference_golden",
        "test_timed_path_gpu.py::test_stage2_headline_sampler_steps_through_graph_and_two_streams_match_reference",
        "test_at_size_gpu.py::test_sampler_25_steps_full_width_matches_reference_golden",
        "test_unet_gpu.py::test_fused_graph_step_equals_generic_step",
        "test_kernels_gpu.py::test_cfg_prepare_and_sampler_step",
        "test_unet_gpu.py::test_guidance_scale_change_after_graph_capture_takes_effect"]),
    ("a7-a15 VideoUNet.forward and its blocks", [
        "test_unet_gpu.py::test_unet_matches_reference_golden",
        "test_unet_gpu.py::test_unet_full_size_stage1_matches_reference_golden",
        "test_at_size_gpu.py::test_unet_full_size_stage2_matches_reference_golden[bf16]",
        "test_at_size_gpu.py::test_unet_full_width_32_views_matches_reference_golden",
        "test_at_size_gpu.py::test_unet_config4_full_size_matches_reference_golden",
        "test_unet_gpu.py::test_unet_vs_oracle_other_shape",
        "test_unet_gpu.py::test_unet_32_views_vs_oracle"]),
    ("a16 decode_first_stage / Decoder", [
        "test_vae_gpu.py::test_vae_decode_matches_reference_golden",
        "test_at_size_gpu.py::test_vae_decode_full_resolution_matches_reference_golden",
        "test_vae_gpu.py::test_engine_decode_first_stage_chunks"]),
    ("a17 encode_first_stage / Encoder / posterior", [
        "test_vae_gpu.py::test_vae_encode_matches_reference_golden",
        "test_at_size_gpu.py::test_vae_encode_full_resolution_matches_reference_golden",
        "test_vae_gpu.py::test_autoencoding_engine_encode_matches_reference_golden",
        "test_vae_gpu.py::test_cond_frame_embedder_matches_reference_encoder"]),
    ("a18 v02 blend loop", [
        "test_unet_gpu.py::test_stage2_refine_loop_matches_reference_golden",
        "test_at_size_gpu.py::test_stage2_refine_25_steps_full_width_matches_reference_golden",
        "test_timed_path_gpu.py::test_stage2_refine_loop_full_size_25_steps_and_decode_match_reference_end_to_end"]),
    ("a20 / f1 VideoDecoder, AE3DConv", ["test_vae_gpu.py::test_video_decoder"]),
    ("b boundary: torch.ops.hi3d, error convention", ["test_torch_ops_gpu.py::", "test_kernels_gpu.py::test_errors_are_loud"]),
    ("e multi-GPU: CFG split, frame<->space all-to-all, sharded decode", ["test_parallel_gpu.py::"]),
    ("f2 VAE encoder + v02 pre-loop, clips end to end", [
        "test_depth_gpu.py::test_v02_conditioner_end_to_end", "test_pipeline_gpu.py::test_stage2_clip_from_yaml",
        "test_pipeline_gpu.py::test_stage1_clip_create_model_sample_decode",
        "test_timed_path_gpu.py::test_stage1_clip_full_size_25_steps_and_decode_match_reference_end_to_end",
        "test_timed_path_gpu.py::test_full_width_25_step_latents_decode_to_the_reference_images"]),
    ("f3 conditioner on the GPU", [
        "test_clip_gpu.py::test_openclip_prediction_embedder_end_to_end", "test_clip_gpu.py::test_aes_embedder_end_to_end",
        "test_clip_gpu.py::test_vit_runtime", "test_depth_gpu.py::test_depth_embedder", "test_depth_gpu.py::test_dpt_hybrid_matches_reference_midas"]),
    ("f4 checkpoint loader, re-layout cache", [
        "test_pipeline_gpu.py::test_deepspeed_checkpoint_to_runtime_matches_reference_golden",
        "test_pipeline_gpu.py::test_pack_cache_cold_and_warm_are_bit_identical"]),
    ("N1 fp8 attention (BASELINE config 5)", [
        "test_at_size_gpu.py::test_unet_full_size_stage2_matches_reference_golden[fp8",
        "test_at_size_gpu.py::test_sampler_25_steps_full_width_fp8_attention", "test_unet_gpu.py::test_unet_fp8_attention_paths",
        "test_timed_path_gpu.py::test_stage2_headline_sampler_steps_fp8_attention_match_reference"]),
]
# run last: timing-stress / race screens / long repeatability runs, and tests of code paths that are OFF by default
LAST_TESTS = ["isa_timing_stress", "race_screen", "ragged_repeatable", "test_groupnorm_folded_into_the_linear_layer",
              "test_attention_d512", "test_gemm_split_k_scratch_is_per_stream"]


def row_priority(nodeid):
    """(class, index): class 0 = evidence of a SURVEY 8 row (index = position in ROW_TESTS), 1 = everything else, 2 = LAST_TESTS."""
    for i, (_, pats) in enumerate(ROW_TESTS):
        if any(p in nodeid for p in pats):
            return (0, i)
    if any(p in nodeid for p in LAST_TESTS):
        return (2, 0)
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    order = {id(it): n for n, it in enumerate(items)}
    items.sort(key=lambda it: row_priority(it.nodeid) + (order[id(it)],))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _gemm_env_follows_monkeypatch(request, monkeypatch):
    """The GEMM dispatch caches its HI3D_GEMM_* switches per process (hi3d_gemm_reload_env).  Tests flip them with
    monkeypatch.setenv: this wrapper makes setenv / delenv of such a name re-read the cache at once, and re-reads it again after
    the test, when monkeypatch has restored the environment (this fixture depends on monkeypatch, so it is torn down FIRST --
    hence the explicit undo here before the final reload)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from hi3d_hip import ops
    names = ("HI3D_GEMM_", "HI3D_GN_FUSED_OFF", "HI3D_GN_POST", "HI3D_CONVT_SKIP")
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_(name, value, *a, **k)
        if name.startswith(names):
            ops.gemm_reload_env()

    def delenv(name, *a, **k):
        del_(name, *a, **k)
        if name.startswith(names):
            ops.gemm_reload_env()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    ops.gemm_reload_env()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def shrink_conditioner(y):
    """Reduced CLIP towers in a loaded inference YAML (2 layers instead of 32 / 24: the full ones hold 0.94 B parameters);
    same classes, key names and output widths."""
    for e in y["model"]["params"]["conditioner_config"]["params"]["emb_models"]:
        if e["target"].endswith("FrozenOpenCLIPImagePredictionEmbedder"):
            e["params"]["open_clip_embedding_config"]["params"]["arch"] = "ViT-tiny-H"
        elif e["target"].endswith("AesEmbedder"):
            e.setdefault("params", {})["arch"] = "ViT-tiny-L"
    return y


_SYNTH_WEIGHTS = {}         # (state_dict key, shape, seed, device) -> fp32 tensor on the device: drawn once per session


def synth_fill_cached(module, key_prefix, seed, dev):
    """Load the hi3d_hip.synth weights (the CPU stream: the values the reference modules got when a golden was made) of the
    keys `key_prefix + k` into `module`, which already lives on `dev`.  The drawn tensors are kept on the device for the session,
    per TENSOR -- the stage-1 / stage-2 networks (same keys but the input conv) share them -- because drawing 1.5 B parameters on
    the host was half of the GPU suite's wall time."""
    from hi3d_hip import synth
    sd = {}
    for k, v in module.state_dict().items():
        if v.dtype.is_floating_point:
            ck = (key_prefix + k, tuple(v.shape), seed, str(dev))
            if ck not in _SYNTH_WEIGHTS:
                _SYNTH_WEIGHTS[ck] = synth.synth_tensor(ck[0], v.shape, seed).to(dev)
            sd[k] = _SYNTH_WEIGHTS[ck]
    module.load_state_dict(sd, strict=False)
    return module


def synth_unet(fx, dev):
    """A fresh VideoUNet(**fx["cfg"]) on `dev` carrying the fixture's seeded weights.  Every test gets its OWN module (its own
    runtime, built under the test's environment); what is shared for the session is the drawn state dict (synth_fill_cached);
    torch's default init of a module that is overwritten anyway is skipped."""
    import torch
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.util import ParamTree
    ParamTree.skip_init = True
    try:
        with torch.device(dev):
            m = VideoUNet(**fx["cfg"])
    finally:
        ParamTree.skip_init = False
    return synth_fill_cached(m, fx["key_prefix"], fx["weight_seed"], dev)
