"""world_size-2 gloo tests (CPU) of the multi-GPU mappings: VAE frame sharding +
all-gather, the CFG-pair split, and the frame <-> space re-sharding of ONE clip inside the UNet
 -- the last one with the real network arithmetic: the CPU oracle UNet runs on every
rank under hi3d_hip.parallel.FrameSpaceGroup and must reproduce the unsharded oracle.  (The VAE / CFG
tests use an analytic stand-in network: what they check is the collective logic.)"""
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
    sys.path.insert(0, root)
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


# ---------------------------------------------------------------- frame <-> space all-to-all
def _reshard_job(rank, world):
    """layout algebra of the two switches on a labelled tensor: element (b, t, s, c) must land where the
    layouts say, and the round trip must be the identity"""
    from hi3d_hip.parallel import FrameSpaceGroup
    out = {}
    for B, T, S, C in ((1, 4, 8, 3), (2, 8, 12, 2)):
        g = FrameSpaceGroup(T)
        Tl, Sl = T // world, S // world
        full = torch.arange(B * T * S * C, dtype=torch.float32).reshape(B, T, S, C)
        mine_f = full[:, rank * Tl:(rank + 1) * Tl].reshape(B * Tl * S, C)            # frame-sharded
        mine_s = full[:, :, rank * Sl:(rank + 1) * Sl].reshape(B * T * Sl, C)         # space-sharded
        got_s = g.frames_to_space(mine_f.clone(), B, S)
        got_f = g.space_to_frames(mine_s.clone(), B, S)
        back = g.space_to_frames(got_s.clone(), B, S)
        out[(B, T)] = (torch.equal(got_s, mine_s), torch.equal(got_f, mine_f), torch.equal(back, mine_f),
                       g.n_switches, g.bytes_moved, list(g.local_frames(B)))
    return out


def test_frame_space_all_to_all_layouts():
    for world in (2, 4):
        res = spawn(_reshard_job, world)
        for r in range(world):
            for (B, T), (a, b, c, n, nbytes, lf) in res[r].items():
                assert a and b and c and n == 3
                Tl = T // world
                assert lf == [bb * T + r * Tl + i for bb in range(B) for i in range(Tl)]
        # bytes: every switch ships (w-1)/w of the local shard
        B, T, S, C = 2, 8, 12, 2
        assert res[0][(2, 8)][4] == 3 * (B * T * S * C * 4 // world) * (world - 1) // world


def _sharded_case(world):
    """(T, H, W, B): every level's pixel count and T must divide by the frame-parallel degree -- 2 ranks: the round-2 case;
    8 ranks (one whole MI355X node as cfg 1 x sp 8): 8 frames, 32 x 32 latents (4 x 4 = 16 pixels at the lowest level)"""
    return (4, 16, 8, 2) if world <= 2 else (8, 32, 32, 2)


def _sharded_inputs(world):
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_s2_ioi.pt"), weights_only=False)
    from hi3d_hip import synth
    cfg, P = fx["cfg"], fx["key_prefix"]
    sd = synth.synth_state_dict({P + k: v for k, v in fx["shapes"].items()}, fx["weight_seed"])
    T, H, W, B = _sharded_case(world)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B * T, cfg["in_channels"], H, W), generator=g)
    ctx, y = torch.randn((B, 1, 1024), generator=g), torch.randn((B, cfg["adm_in_channels"]), generator=g)
    ioi = torch.zeros(B, T); ioi[1, 2] = 1.0
    return cfg, P, sd, x, ctx, y, ioi, T, B


def _sharded_unet_job(rank, world):
    from hi3d_hip.parallel import FrameSpaceGroup
    from oracle import hi3d_oracle_sharded as OS
    cfg, P, sd, x, ctx, y, ioi, T, B = _sharded_inputs(world)
    comm = FrameSpaceGroup(T)
    torch.set_num_threads(2 if world <= 2 else 1)
    with torch.no_grad():
        out = OS.video_unet_sharded(sd, cfg, x[comm.local_frames(B)], 0.37, ctx, y, T, ioi, comm, prefix=P)
    return out, comm.n_switches, comm.n_allreduce, comm.bytes_moved


def _sharded_unet_two_comm_job(rank, world):
    """cfg 1 x sp `world` with the OVERLAP decomposition of ClipParallelStepper(overlap=True): a rank holds both CFG halves of
    its frames and runs them as two independent chains, each on its OWN communicator over the same ranks (B = 1 per chain:
    one half's exchange can then run beside the other half's spatial sub-block).  Here the chains run one after the other
    (gloo is blocking); what is checked is the decomposition: two B = 1 passes on two process groups == the B = 2 pass."""
    from hi3d_hip.parallel import FrameSpaceGroup
    from oracle import hi3d_oracle_sharded as OS
    cfg, P, sd, x, ctx, y, ioi, T, B = _sharded_inputs(world)
    ranks = list(range(world))
    comms = [FrameSpaceGroup(T, dist.new_group(ranks)) for _ in range(B)]
    torch.set_num_threads(1)
    outs = []
    with torch.no_grad():
        for b in (1, 0):                        # (the side chain is issued first, as forward_tokens does)
            xb = x[b * T:(b + 1) * T][comms[b].local_frames(1)]
            outs.append((b, OS.video_unet_sharded(sd, cfg, xb, 0.37, ctx[b:b + 1], y[b:b + 1], T, ioi[b:b + 1], comms[b], prefix=P)))
    outs.sort(key=lambda t: t[0])
    return torch.cat([o for _, o in outs], 0), [c.n_switches for c in comms], [c.n_allreduce for c in comms], sum(c.bytes_moved for c in comms)


def _check_sharded(res, world, two_comm=False):
    from hi3d_hip.runtime_unet import unet_layout
    from oracle import hi3d_oracle as O
    cfg, P, sd, x, ctx, y, ioi, T, B = _sharded_inputs(world)
    with torch.no_grad():
        ref = O.video_unet(sd, cfg, x, torch.full((B * T,), 0.37), ctx, y, T, ioi, prefix=P)
    Tl = T // world
    got = torch.empty_like(ref)
    for r in range(world):
        idx = torch.cat([torch.arange(b * T + r * Tl, b * T + (r + 1) * Tl) for b in range(B)])
        got[idx] = res[r][0]
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"frame<->space sharded oracle vs unsharded, {world} ranks{' (one communicator per CFG half)' if two_comm else ''}: rel {err:.2e}")
    assert err < 2e-5
    # the plan: two switches per VideoResBlock and per SpatialVideoTransformer, two sum all-reduces per 3-D ResBlock
    bi, mid, bo = unet_layout(cfg)
    layers = [L for blk in bi + [mid] + bo for L in blk]
    nres, nattn = sum(L[0] == "res" for L in layers), sum(L[0] == "attn" for L in layers)
    if two_comm:
        assert res[0][1] == [2 * (nres + nattn)] * B and res[0][2] == [2 * nres] * B
    else:
        assert res[0][1] == 2 * (nres + nattn) and res[0][2] == 2 * nres
    return res[0][3]


@pytest.mark.parametrize("world", [2, 8])
def test_unet_frame_space_sharded_equals_unsharded_oracle(world):
    """`world` ranks, each holding 1/world of the frames (spatial sub-blocks) / of the pixels (temporal sub-blocks) of the same
    2-clip batch, with the 3-D GroupNorm partial-sum all-reduce: the concatenation of the ranks' outputs must equal the
    single-process oracle forward.  8 ranks = one whole node as cfg 1 x sp 8."""
    _check_sharded(spawn(_sharded_unet_job, world), world)


def test_unet_two_communicators_one_per_cfg_half_8_ranks():
    """The decomposition behind the all-to-all / compute overlap of the cfg 1 x sp 8 mapping: the two CFG halves as two B = 1
    chains on two communicators over the same 8 ranks reproduce the unsharded oracle, move the same bytes as the joint B = 2
    pass and issue the same number of collectives per chain."""
    world = 8
    moved_two = _check_sharded(spawn(_sharded_unet_two_comm_job, world), world, two_comm=True)
    moved_joint = spawn(_sharded_unet_job, world)[0][3]
    assert moved_two == moved_joint


def _groups_job(rank, world):
    """cfg x sp sub-groups of the clip-parallel mapping: on the whole job (4 ranks = cfg 2 x sp 2), and with TWO clips side
    by side (parents {0,1} and {2,3}, each cfg 1 x sp 2) -- the parents and the sub-groups are created by their members
    only (group-local synchronisation), the way two steppers on disjoint GPUs of one job do it."""
    from hi3d_hip.parallel import clip_parallel_groups
    out = {}
    sp, half, part, g = clip_parallel_groups(None, cfg=2)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, group=g)
    out["whole"] = (sp, half, part, dist.get_process_group_ranks(g), t.item())
    sp2, half2, part2, g2 = clip_parallel_groups(None, cfg=2)              # cached: the same communicator
    out["cached"] = g2 is g
    members = [0, 1] if rank < 2 else [2, 3]
    parent = dist.new_group(members, use_local_synchronization=True)
    sp, half, part, g = clip_parallel_groups(parent, cfg=1)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, group=g)
    out["two_clips"] = (sp, half, part, dist.get_process_group_ranks(g), t.item())
    sp, half, part, g = clip_parallel_groups(parent, cfg=2)               # sp = 1: a group of one
    out["cfg_only"] = (sp, half, part, dist.get_process_group_ranks(g))
    return out


def test_clip_parallel_group_membership_4_ranks():
    res = spawn(_groups_job, 4)
    for r in range(4):
        half, part = divmod(r, 2)
        mem = [2 * half, 2 * half + 1]
        assert res[r]["whole"] == (2, half, part, mem, float(sum(mem)))
        assert res[r]["cached"]
        assert res[r]["two_clips"] == (2, 0, r % 2, mem, float(sum(mem)))
        assert res[r]["cfg_only"] == (1, r % 2, 0, [r])
