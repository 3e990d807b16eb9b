"""world_size-2 gloo tests (CPU) of the multi-GPU mappings: VAE frame sharding +
all-gather, and the CFG-pair split.  The arithmetic under test is the sharding /
collective logic; the per-rank 'network' is an analytic stand-in (kernels need a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "hi3d-official_amd"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _fake_decode(z):                      # per-frame, like the 2-D VAE: frame f only depends on z[f]
    return torch.nn.functional.interpolate(torch.tanh(z[:, :3]) * 2 + z[:, 3:4], scale_factor=2.0)


def _decode_job(rank, world):
    from hi3d_hip.parallel import decode_sharded, frame_slice
    outs = {}
    for T in (16, 5, 1):                   # even split, ragged split, fewer frames than ranks
        z = torch.randn((T, 4, 6, 6), generator=torch.Generator().manual_seed(T))
        outs[T] = decode_sharded(_fake_decode, z)
        lo, hi = frame_slice(T, rank, world)
        assert 0 <= lo <= hi <= T
    return outs


def test_vae_frame_shard_allgather_matches_single_process():
    res = spawn(_decode_job, 2)
    for T in (16, 5, 1):
        z = torch.randn((T, 4, 6, 6), generator=torch.Generator().manual_seed(T))
        ref = _fake_decode(z)
        for r in range(2):
            assert torch.equal(res[r][T], ref)


def _cfg_job(rank, world):
    import sys
    from hi3d_hip.parallel import SplitCFGGuider
    from hi3d_hip import synth
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    T, steps = 4, 5
    sampler = EulerEDMSampler(
        num_steps=steps, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}})
    sampler.guider = SplitCFGGuider(sampler.guider)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    x0, c, uc = synth.synth_conditioning(T, 6, 6, stage=1, seed=3)

    def network(x, c_noise, cond, **kw):
        return torch.tanh(x) * c_noise.reshape(-1, 1, 1, 1) + cond["concat"].mean(1, keepdim=True) + cond["vector"].mean()

    return sampler(lambda i, s, cc: den(network, i, s, cc), x0.clone(), cond=c, uc=uc)


def test_cfg_pair_split_matches_doubled_batch():
    from hi3d_hip import synth
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    res = spawn(_cfg_job, 2)
    T, steps = 4, 5
    sampler = EulerEDMSampler(
        num_steps=steps, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}})
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    x0, c, uc = synth.synth_conditioning(T, 6, 6, stage=1, seed=3)

    def network(x, c_noise, cond, **kw):
        return torch.tanh(x) * c_noise.reshape(-1, 1, 1, 1) + cond["concat"].mean(1, keepdim=True) + cond["vector"].mean()

    ref = sampler(lambda i, s, cc: den(network, i, s, cc), x0.clone(), cond=c, uc=uc)
    for r in range(2):
        assert torch.allclose(res[r], ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(res[0], res[1])
