#!/usr/bin/env python
"""Benchmark of the Hi3D denoising hot path on MI355X.

metric : denoise-steps/sec (one step = one EulerEDMSampler.sampler_step = CFG-doubled
         VideoUNet forward at batch 2*T + guider + Euler update), BASELINE.json.
workload (N=1): stage-2 refiner, 16 views @ 1024x1024 (latent 128x128, in_channels 17),
         bf16 storage / fp32 accumulate, random-init weights of the full 1.52 B-parameter
         architecture (hi3d_hip.synth), synthetic conditioning, inputs resident in HBM.
         `--config s1` selects stage-1 (16 views @ 512x512) instead; `--config vae` times the
         first-stage decode of the clip (decode_first_stage, 16 frames @ 1024x1024) as its own line.
timing : the K timed steps run the product path -- one HIP-graph replay per step
         (hi3d_hip/fused_step.py).  Kernels inside a graph cannot be bracketed by events, so the
         per-kernel HIP-event breakdown (`roofline`, `kernels_ms_per_step`) is taken over the K steps
         that FOLLOW the timed region, same inputs and same kernels launched eagerly on the same stream.
legs   : with the default arguments at N = 1 the headline line also carries `legs` -- the other BASELINE.json
         configurations measured by the same code on the same box, each with its own executed-FLOP roofline:
         stage 1 (config 2: 16 views @ 512^2), stage 2 at 32 views (config 4), stage 2 with the fp8 score product
         (config 5) and the first-stage decode of the clip.  `--no-legs` skips them.
N > 1  : one process per GPU (torchrun); every rank denoises an independent orbit clip
         (replicas -- the unit that shards with no data-path collective);
         value = steps of all ranks / max-over-ranks time; "scaling": "weak".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def unet_cfg(stage, mc=320):
    return dict(in_channels=8 if stage == 1 else 17, model_channels=mc, out_channels=4, num_res_blocks=2,
                attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_head_channels=64,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
                spatial_transformer_attn_type="softmax-xformers", extra_ff_mix_layer=True, use_spatial_context=True,
                merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1], num_classes="sequential",
                adm_in_channels=768 if stage == 1 else 512, use_checkpoint=True)


# algorithmic work per denoise step of the reference-equivalent graph, and the part of it this
# framework executes: the single-key cross-attention (softmax == 1) is eliminated algebraically, SURVEY 8d asks for
# utilisation against the EXECUTED work when that is the case.  The executed figure is counted per launch by the
# profiler (ops.Profiler: 2 M N K per GEMM / conv, 4 b h n m d per attention); these constants are the same count,
# used only when the per-kernel profile is switched off.
STEP_TFLOP = {1: 40.61, 2: 209.47}
STEP_TFLOP_EXECUTED = {1: 37.7, 2: 197.83}    # (round 4: the up-sampling convs run as four 2x2 phase convs, 4/9 of their multiply-adds)
PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0       # dense MFMA fp8 (MX-scaled K = 128 forms), same guide
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The contract is ONE JSON line on stdout.  Libraries write to fd 1 behind Python's back -- RCCL prints a five-line version banner
# there when the first communicator is made (seen in the one-rank dry run of the multi-GPU legs) -- so main() takes the real
# stdout aside and points fd 1 at stderr for the rest of the process; emit() is the only writer of the real one.
_REAL_STDOUT = None


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def guard_line(line):
    """Insurance for the ONE JSON line while an optional leg runs that could take the process down hard (a collective library
    aborting cannot be caught in Python): a detached helper that inherits stdout blocks on a pipe whose write end only THIS
    process holds.  disarm() sends it a byte (the normal path: the caller prints the complete line itself); if the process
    dies instead, the helper sees end-of-file and prints `line`."""
    import subprocess
    r, w = os.pipe()
    code = ("import os,sys\n"
            "b=os.read(int(sys.argv[1]),1)\n"
            "if b==b'':\n    sys.stdout.write(sys.argv[2]+'\\n'); sys.stdout.flush()\n")
    subprocess.Popen([sys.executable, "-c", code, str(r), line], stdin=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                     stdout=_REAL_STDOUT,          # (None before claim_stdout(): inherit fd 1)
                     start_new_session=True, pass_fds=(r,))
    os.close(r)
    state = {"armed": True}

    def disarm():
        if state["armed"]:
            state["armed"] = False
            os.write(w, b"x")
            os.close(w)
    return disarm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["s1", "s2", "vae"], default="s2")
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--attn", choices=["bf16", "fp8qk", "fp8"], default="bf16",
                    help="BASELINE config 5 (fp8 MFMA attention + bf16 conv; reduced precision, own tolerances): fp8qk = the score "
                         "product of the spatial attention on the e4m3 / e8m0-scaled matrix path, P V bf16; fp8 = both products")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--shapes", action="store_true", help="also log the per-shape breakdown of the profiled step")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N = 1 headline run: do not re-measure roofline.traffic with two rocprofv3 --pmc passes of a child run (the "
                         "committed profiles/traffic_s2.json is quoted instead, labelled)")
    ap.add_argument("--no-legs", action="store_true", help="N = 1 headline run: skip the extra legs (stage 1, 32 views, fp8 scores, VAE decode)")
    ap.add_argument("--leg-steps", type=int, default=6)
    ap.add_argument("--simulate-sp", type=int, default=4,
                    help="N = 1: also time what ONE rank of a cfg2 x spN clip-parallel mapping computes per step (per-rank shapes, pack / "
                         "unpack kernels, no peers: hi3d_hip.parallel.SimulatedFrameSpaceGroup); 0 = skip")
    ap.add_argument("--model-coll-us", type=float, default=25.0,
                    help="assumed launch + rendezvous latency of ONE RCCL collective inside the step, for the modelled exchange time of the "
                         "simulated rank (an assumption printed with the result, not a measurement)")
    ap.add_argument("--no-clip-parallel", action="store_true",
                    help="N > 1: skip the extra leg that runs ONE clip over all GPUs (CFG split x frame<->space all-to-all, RCCL)")
    a = ap.parse_args()
    claim_stdout()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {a.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in this framework)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torchrun
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.config == "vae":
        res = bench_vae(a, rank, world, dev, use_dist, a.steps, a.warmup)
        if rank == 0:
            emit(json.dumps(res))
        if use_dist:
            torch.distributed.destroy_process_group()
        return
    stage = 1 if a.config == "s1" else 2
    out, unet, sampler, ms_per_step = unet_bench(a, stage, a.views, a.attn, rank, world, dev, use_dist, a.steps, a.warmup,
                                                 profile=not a.no_profile, cpu=(rank == 0 and world == 1 and not a.no_cpu_baseline))
    headline = stage == 2 and a.views == 16 and a.attn == "bf16"
    if headline and world == 1 and not use_dist and not a.no_legs:
        sim = None
        if a.simulate_sp > 1:
            try:
                sim = simulate_sp_leg(a, unet, a.views, 128, dev, a.simulate_sp, ms_per_step)
            except Exception as e:                      # noqa: BLE001
                sim = {"error": f"{type(e).__name__}: {e}"}
        del unet, sampler
        out["legs"] = run_legs(a, rank, world, dev)
        if sim is not None:
            out["legs"][f"clip_parallel_cfg2_sp{a.simulate_sp}_one_rank_simulated"] = sim
        unet = sampler = None
        if not a.no_traffic and not a.no_profile and "roofline" in out:
            gc_ = __import__("gc"); gc_.collect(); torch.cuda.empty_cache()
            two = "two HIP streams" in out["config"].get("streams", "")
            # (the headline and the legs are measured at this point -- a helper prints the line as it stands if this
            # process is killed while the two PMC child runs are under way, e.g. by a time-limited driver)
            try:
                disarm = guard_line(json.dumps(dict(out, note="printed by the line guard: the process ended during the live traffic passes")))
            except Exception:                           # noqa: BLE001
                disarm = lambda: None
            try:
                live, src = measure_traffic_live(ln_rows=2 * a.views * 128 * 128 // (2 if two else 1))
            except Exception as e:                      # noqa: BLE001 -- never lose the line to the optional measurement
                live, src = None, f"{type(e).__name__}: {e}"
            disarm()
            kern = out["roofline"]["kernel"]
            if live and kern in live:
                out["roofline"]["traffic"] = live[kern]["bytes_per_launch"]
                ab = out["roofline"].get("algorithmic_bytes_per_launch")
                if ab:
                    out["roofline"]["traffic_over_algorithmic"] = round(live[kern]["bytes_per_launch"] / ab, 3)
                out["roofline"]["traffic_source"] = src
                out["roofline"]["traffic_detail"] = live
            else:
                out["roofline"]["traffic_source"] = ("committed profiles/traffic_s2.json (the builder's PMC passes of the same command); "
                                                     f"live measurement unavailable: {src}")
    # (HI3D_BENCH_FORCE_MULTI=1: run the multi-GPU legs in a ONE-rank nccl group too -- a dry run of their code on a one-GPU box)
    if use_dist and (world > 1 or os.environ.get("HI3D_BENCH_FORCE_MULTI") == "1") and not a.no_clip_parallel:
        # second leg (not `value`): the SAME step for ONE clip spread over all GPUs -- the mapping that makes a
        # single 16/32-view clip faster: CFG pair x frame<->space groups, all collectives over RCCL
        # ... and never lose the headline line to it: a Python error is caught, and a watchdog prints the line and ends the
        # process if the leg does not come back (a stuck collective cannot be interrupted from Python)
        import threading

        LEGS = ("clip_parallel", "clip_parallel_32views", "vae_decode_sharded", "clip_parallel_cfg1_overlap")
        state = {"leg": LEGS[0], "disarm": (lambda: None)}

        def _bail():
            state["disarm"]()
            if rank == 0:      # (only the leg that did not come back is marked: the finished ones keep their results)
                out[state["leg"]] = {"error": "timed out (watchdog); the headline numbers and the finished legs are unaffected"}
                emit(json.dumps(out))
            os._exit(0)

        def entering(leg):
            """Before each leg: the watchdog restarts with that leg's name, and the detached helper that prints the line if THIS
            process dies (an abort inside the collective library) is re-armed with everything measured so far."""
            state["leg"] = leg
            state["disarm"]()
            state["disarm"] = lambda: None
            if state.get("wd") is not None:
                state["wd"].cancel()
            if rank == 0:
                try:
                    state["disarm"] = guard_line(json.dumps(dict(out, **{leg: {"error": "process died inside this optional leg; the "
                                                                                 "headline numbers and the finished legs are unaffected"}})))
                except Exception as e:                    # noqa: BLE001 -- the guard is insurance, never a reason to fail
                    log(f"[bench] line guard not armed: {type(e).__name__}: {e}")
            state["wd"] = threading.Timer(max(240.0, 120.0 * (a.steps + a.warmup + 2) * ms_per_step / 1e3), _bail)
            state["wd"].daemon = True
            state["wd"].start()

        entering("clip_parallel")
        lat = 64 if stage == 1 else 128
        try:
            out["clip_parallel"] = clip_parallel_leg(a, unet, sampler.guider, stage, a.views, lat, dev, world, ms_per_step)
        except Exception as e:
            out["clip_parallel"] = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE config 4: "Stage-2 32 views @ 1024^2, view-parallel shard over 8 x MI355X with RCCL all-gather at VAE decode":
        # the 32-view clip on all GPUs (CFG pair x frame<->space groups), and the sharded decode of its frames + the all-gather
        if "error" not in out["clip_parallel"] and stage == 2 and a.views == 16:
            entering("clip_parallel_32views")
            try:
                from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider
                g32 = LinearPredictionGuider(max_scale=2.0, num_frames=32, min_scale=1.0)
                out["clip_parallel_32views"] = clip_parallel_leg(a, unet, g32, stage, 32, lat, dev, world, None,
                                                                 steps=max(2, a.steps // 2))
            except Exception as e:
                out["clip_parallel_32views"] = {"error": f"{type(e).__name__}: {e}"}
        if not any("error" in out.get(k, {}) for k in ("clip_parallel", "clip_parallel_32views")):
            entering("vae_decode_sharded")
            torch.cuda.empty_cache()
            try:
                out["vae_decode_sharded"] = vae_sharded_leg(a, 32 if (stage == 2 and a.views == 16) else a.views, lat, dev, world)
            except Exception as e:
                out["vae_decode_sharded"] = {"error": f"{type(e).__name__}: {e}"}
        # LAST (it has never run over RCCL: whatever it does, the legs above are already in the line):
        # the other mapping of the same clip: cfg 1 x sp N -- every rank holds BOTH CFG halves of its frames and runs them as two
        # chains on two streams with a communicator each, so one half's exchange overlaps the other half's compute (the cfg 2
        # mapping above has no independent work on a rank to overlap with); more, smaller collectives, no idle CUs while waiting
        if not any("error" in out.get(k, {}) for k in ("clip_parallel", "clip_parallel_32views", "vae_decode_sharded")):
            entering("clip_parallel_cfg1_overlap")
            try:
                out["clip_parallel_cfg1_overlap"] = clip_parallel_leg(a, unet, sampler.guider, stage, a.views, lat, dev, world, ms_per_step,
                                                                      steps=max(2, a.steps // 2), cfg_split=1, overlap=True)
            except Exception as e:
                out["clip_parallel_cfg1_overlap"] = {"error": f"{type(e).__name__}: {e}"}
        state["wd"].cancel()
        state["disarm"]()
    if rank == 0:
        emit(json.dumps(out))
    if use_dist:
        if any("error" in out.get(k, {}) for k in ("clip_parallel", "clip_parallel_cfg1_overlap", "clip_parallel_32views", "vae_decode_sharded")):
            sys.stdout.flush()
            os._exit(0)                  # ranks may have diverged inside the optional leg: do not wait on a teardown barrier
        torch.distributed.destroy_process_group()


def measure_traffic_live(ln_rows):
    """HBM-side traffic per launch of the dominant kernels, MEASURED in this run: two rocprofv3 --pmc passes (FETCH_SIZE, then
    WRITE_SIZE -- separate runs, kernel trace only, as MI355X_MICROARCH.md's HBM section and the pool's rules prescribe) over a
    child `bench.py --steps 1 --warmup 1 --no-legs --no-profile` with HI3D_STEP_GRAPH=0 (4 eager sampler steps and nothing
    else: every profiled launch belongs to a step, same launch population as the per-kernel profile pass).  FETCH_SIZE is
    doubled (gfx950 counts 64 B per 128-B request for 16 B/lane streams); both counters are in KB.  Calibration in the same
    pass: the C = 320 LayerNorm reads ln_rows x 640 B per launch.  Returns (dict or None, source string)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    fams = ("gemm_bf16_kernel", "attn_d64_kernel", "ffn2_geglu_c320_kernel", "layernorm_packed_kernel<8>")
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hi3d_pmc_", dir="/tmp")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        env.update(HI3D_STEP_GRAPH="0", HI3D_BENCH_PARITY="0", TMPDIR="/tmp")
        cmd = [rp, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "run", "--output-format", "csv", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-legs", "--no-traffic"]
        t0 = time.time()
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=180)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)            # (the group THIS call started, nothing else)
                pr.wait()
                shutil.rmtree(d, ignore_errors=True)
                return None, f"{counter} pass timed out"
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if rc != 0 or not files:
                shutil.rmtree(d, ignore_errors=True)
                return None, f"{counter} pass failed (rc {rc}, {len(files)} csv)"
            agg = {f: [0, 0.0] for f in fams}
            with open(files[0]) as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] != counter:
                        continue
                    for f in fams:
                        if f in r["Kernel_Name"]:
                            agg[f][0] += 1; agg[f][1] += float(r["Counter_Value"])
                            break
            tot[counter] = agg
            log(f"[bench] live PMC pass {counter}: {time.time() - t0:.0f}s, {sum(v[0] for v in agg.values())} launches of the tracked kernels")
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res = {}
    for f in fams[:3]:
        n = max(tot["FETCH_SIZE"][f][0], tot["WRITE_SIZE"][f][0])
        if n:
            res[f] = {"bytes_per_launch": int((2.0 * tot["FETCH_SIZE"][f][1] + tot["WRITE_SIZE"][f][1]) * 1024 / n), "launches_profiled": n,
                      "fetch_KB_x2_corrected": int(2.0 * tot["FETCH_SIZE"][f][1]), "write_KB": int(tot["WRITE_SIZE"][f][1])}
    ln = tot["FETCH_SIZE"][fams[3]]
    if ln[0]:
        res["calibration"] = {"kernel": "layernorm_packed_kernel<8> (C = 320)", "fetch_KB_per_launch_reported": round(ln[1] / ln[0], 1),
                              "KB_actually_read_per_launch": ln_rows * 640 // 1024,
                              "x2_corrected_over_actual": round(2.0 * ln[1] / ln[0] / (ln_rows * 640 / 1024), 4)}
    return res, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench.py run (FETCH_SIZE x 2 + WRITE_SIZE, KB -> B)"


def simulate_sp_leg(a, unet, T, lat, dev, sp, single_gpu_ms):
    """Compute side of the clip-parallel mapping without a multi-GPU node: what ONE rank of `cfg2 x sp` runs per step --
    one CFG half, T / sp frames in the spatial sub-blocks and S / sp pixels in the temporal ones, the pack / unpack kernels
    around every (absent) all-to-all -- timed on this GPU and compared with 1 / (2 sp) of the single-GPU step."""
    from hi3d_hip import ops
    from hi3d_hip.parallel import SimulatedFrameSpaceGroup
    from hi3d_hip.runtime_unet import CIN_PAD
    if T % sp or (lat // 8) ** 2 % sp:
        return {"skipped": f"frames ({T}) / lowest-level pixels ({(lat // 8) ** 2}) not divisible by sp={sp}"}
    rt = unet.runtime(dev)
    g = SimulatedFrameSpaceGroup(T, sp)
    HW = lat * lat
    gen = torch.Generator(device=dev).manual_seed(3)
    tok = torch.randn((g.Tl * HW, CIN_PAD), device=dev, generator=gen).to(torch.bfloat16)
    tok[:, unet.cfg["in_channels"]:] = 0
    tvec = torch.full((T,), 0.25 * 1.5, device=dev)
    ctx = torch.randn((1, 1, unet.cfg["context_dim"]), device=dev, generator=gen)
    y = torch.randn((1, unet.cfg["adm_in_channels"]), device=dev, generator=gen)
    with torch.cuda.device(dev), torch.no_grad():
        st = rt.clip_consts(ctx, y, torch.zeros(1, T, device=dev), T, T)
        for _ in range(2):
            o = rt.forward_tokens(tok, T, lat, lat, tvec, st, T, sp=g)
        n0, b0, r0 = g.n_switches, g.bytes_moved, g.n_allreduce
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.leg_steps):
            o = rt.forward_tokens(tok, T, lat, lat, tvec, st, T, sp=g)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.leg_steps * 1e3
        n1, b1, r1 = g.n_switches, g.bytes_moved, g.n_allreduce
        # the same rank step as ONE HIP-graph replay (the single-GPU figure it is compared with is a graph replay
        # too): captured on a stream of its own whose split-K scratch is registered before the capture, as FusedStepper does
        ms_graph = None
        try:
            cap = torch.cuda.Stream(device=dev)
            ops._ensure_gemm_workspace(dev, cap)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=cap):
                o_g = rt.forward_tokens(tok, T, lat, lat, tvec, st, T, sp=g)
            gr.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.leg_steps):
                gr.replay()
            torch.cuda.synchronize()
            ms_graph = (time.perf_counter() - t0) / a.leg_steps * 1e3
            if not torch.isfinite(o_g).all():
                ms_graph = None
            del gr, o_g
        except Exception as e:                      # noqa: BLE001 -- an extra figure, never a reason to lose the leg
            log(f"[bench] simulated rank: graph capture failed: {type(e).__name__}: {e}")
        # where the distance to the ideal goes: the same rank step with HIP events around every kernel (their sum against the
        # eager wall time above = the launch-gap share; measured 37.6 of 38.0 ms: the rank is NOT launch-bound)
        prof = ops.Profiler()
        ops.PROFILER = prof
        try:
            for _ in range(2):
                rt.forward_tokens(tok, T, lat, lat, tvec, st, T, sp=g)
            torch.cuda.synchronize()
        finally:
            ops.PROFILER = None
        fams = {f: round(d["ms"] / 2, 3) for f, d in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"])}
    if not torch.isfinite(o).all():
        raise RuntimeError("non-finite output in the simulated rank")
    ideal = single_gpu_ms / (2 * sp)
    k = a.leg_steps
    # MODELLED exchange (no multi-GPU box here: nothing below is measured): every rank sends (sp - 1) / sp of what it holds, one
    # peer per xGMI link, all links at once (point-to-point fabric: 153 GB/s per link and direction); each collective -- the
    # all-to-alls, the [b, 32, 2] GroupNorm-sum all-reduces, the closing all-gather of the network output -- additionally
    # costs a launch + rendezvous latency that does not overlap with compute in the cfg2 mapping (a rank holds ONE CFG half:
    # the next kernel needs the exchanged tensor).  a.model_coll_us is that latency (assumption, stated in the line).
    n_a2a, n_ar = (n1 - n0) // k, (r1 - r0) // k
    bw_ms = (b1 - b0) / k / (sp - 1) / 153e9 * 1e3
    lat_ms = (n_a2a + n_ar + 1) * a.model_coll_us * 1e-3
    return {"mapping": f"cfg2 x sp{sp}: one of {2 * sp} ranks, peers absent (exchange replaced by a hand-back of the packed buffer)",
            "rank_ms_per_step": round(ms, 2), "single_gpu_ms_per_step": round(single_gpu_ms, 2), "ideal_rank_ms": round(ideal, 2),
            "compute_scaling_efficiency": round(ideal / ms, 3),
            "rank_ms_per_step_graph_replay": None if ms_graph is None else round(ms_graph, 2),
            "compute_scaling_efficiency_graph_replay": None if ms_graph is None else round(ideal / ms_graph, 3),
            "kernels_ms_per_rank_step": fams, "kernels_total_ms": round(sum(fams.values()), 2),
            "all_to_all_per_step": n_a2a, "gn_allreduce_per_step": n_ar,
            "all_to_all_bytes_sent_per_rank_per_step": (b1 - b0) // k,
            "xgmi_time_at_153GBs_per_link_ms": round(bw_ms, 2),
            "modelled_exchange": {"bandwidth_ms": round(bw_ms, 2), "collectives_per_step": n_a2a + n_ar + 1,
                                  "latency_us_per_collective_assumed": a.model_coll_us, "latency_ms": round(lat_ms, 2),
                                  "rank_ms_per_step_with_exchange": round(ms + bw_ms + lat_ms, 2),
                                  "end_to_end_scaling_efficiency_modelled": round(ideal / (ms + bw_ms + lat_ms), 3),
                                  "speedup_of_one_clip_on_%d_gpus_modelled" % (2 * sp): round(single_gpu_ms / (ms + bw_ms + lat_ms), 2),
                                  "note": "MODEL, not a measurement: bytes / (sp - 1) links / 153 GB/s + collectives x assumed latency, "
                                          "no overlap with compute (cfg2: one CFG half per rank, the consumer waits for the exchange)"},
            "note": "rank_ms_per_step = eager launches (the single-GPU figure is a graph replay); communication time is in neither "
                    "figure -- see modelled_exchange for an end-to-end estimate"}


def run_legs(a, rank, world, dev):
    """The other BASELINE.json configurations, measured by the same code right after the headline (N = 1 only).
    Each leg is one dict: ms_per_step, value (steps/s or frames/s), executed TFLOP per step and the roofline on them.
    A failing leg is reported as {"error": ...}; the headline line is already complete at that point."""
    import gc
    legs = {}
    plan = (("s1_16views_512", dict(stage=1, T=16, attn="bf16")),        # BASELINE config 2
            ("s2_32views_1024", dict(stage=2, T=32, attn="bf16")),       # config 4
            ("s2_16views_fp8qk", dict(stage=2, T=16, attn="fp8qk")),     # config 5, score product only (own tolerance)
            ("s2_16views_fp8", dict(stage=2, T=16, attn="fp8")))         # config 5, both attention products on fp8
    for name, kw in plan:
        gc.collect(); torch.cuda.empty_cache()
        try:
            o, u, sm, _ = unet_bench(a, kw["stage"], kw["T"], kw["attn"], rank, world, dev, False, a.leg_steps, 3, profile=True, cpu=False)
            del u, sm
            legs[name] = {k: o[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "step_roofline",
                                            "roofline", "kernels_ms_per_step", "parity_checked", "parity") if k in o}
            legs[name]["workload"] = o["config"]["workload"]
        except Exception as e:                      # noqa: BLE001
            legs[name] = {"error": f"{type(e).__name__}: {e}"}
    gc.collect(); torch.cuda.empty_cache()
    try:
        r = bench_vae(a, rank, world, dev, False, 1, 1)
        legs["vae_decode_16x1024"] = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_frame", "dtype",
                                                        "roofline", "kernels_ms_per_frame")}
    except Exception as e:                          # noqa: BLE001
        legs["vae_decode_16x1024"] = {"error": f"{type(e).__name__}: {e}"}
    return legs


def unet_bench(a, stage, T, attn, rank, world, dev, use_dist, steps, warmup, profile, cpu):
    """One sampler-step benchmark: `warmup` untimed steps, `steps` timed steps of the product path (graph replay), then the
    per-kernel eager profile.  Returns (json dict, unet, sampler, ms_per_step)."""
    from hi3d_hip import ops, synth
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper

    os.environ["HI3D_ATTN_FP8QK"] = "1" if attn == "fp8qk" else "0"       # read when the runtime is built
    os.environ["HI3D_ATTN_FP8"] = "1" if attn == "fp8" else "0"
    lat = 64 if stage == 1 else 128
    cfg = unet_cfg(stage)
    t0 = time.time()
    from sgm.util import ParamTree
    ParamTree.skip_init = True                       # random-init weights are drawn on the device below
    with torch.device(dev):
        unet = VideoUNet(**cfg)
    ParamTree.skip_init = False
    # parity of the timed region itself: where a reference-class fixture of THIS configuration exists
    # (tests/golden/sampler_s*_full_*.pt: the reference's EulerEDMSampler.step_call on the CPU, oracle/gen_golden.py), rank 0
    # carries the fixture's weights (hi3d_hip.synth CPU stream -- drawn on the host, ~20 s) and starts from the fixture's seeded
    # clip, and the first timed steps are compared with the reference after the clock has stopped.  HI3D_BENCH_PARITY=0 skips it.
    parity_fx = None
    fx_name = {(2, 16): "sampler_s2_full_3step", (1, 16): "sampler_s1_full_25step"}.get((stage, T))
    fx_path = os.path.join(ROOT, "tests", "golden", f"{fx_name}.pt")
    if fx_name and attn == "bf16" and rank == 0 and os.path.exists(fx_path) and os.environ.get("HI3D_BENCH_PARITY", "1") != "0":
        parity_fx = torch.load(fx_path, weights_only=False)
        assert (parity_fx["stage"], parity_fx["T"], parity_fx["hw"], parity_fx["steps"]) == (stage, T, lat, 25)
    if parity_fx is not None:
        synth.fill_module_(unet, parity_fx["weight_seed"], prefix=parity_fx["key_prefix"])
    else:
        synth.fill_module_on_device_(unet, seed=1, prefix="model.diffusion_model.")
    cpu_sd = None
    if cpu:
        cpu_sd = {"model.diffusion_model." + k: v.float().cpu() for k, v in unet.state_dict().items()}
    model = OpenAIWrapper(unet)
    unet.runtime(dev)                      # one-time weight re-layout
    log(f"[bench] rank {rank}: model built + packed in {time.time() - t0:.1f}s" + (f" (weights and clip of {fx_name}.pt)" if parity_fx else ""))

    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(
        num_steps=25, device=dev, verbose=False,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 2.5 if stage == 1 else 2.0, "min_scale": 1.0}})
    # a different clip per rank (rank 0: the fixture's clip when the timed steps are parity-checked)
    x0, c, uc = synth.synth_conditioning(T, lat, lat, stage=stage, seed=parity_fx["input_seed"] if parity_fx else rank)
    if parity_fx is not None:
        pr = parity_fx["x0_probe"]
        if not (torch.equal(x0.flatten()[:16], pr["head"]) and abs(float(x0.double().sum()) - pr["sum"]) < 1e-6 * pr["abs_sum"]):
            raise SystemExit("parity fixture: the seeded clip is not the one the reference ran")
    c = {k: v.to(dev) for k, v in c.items()}
    uc = {k: v.to(dev) for k, v in uc.items()}
    extra = dict(image_only_indicator=torch.zeros(2, T, device=dev), num_video_frames=T)

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(x0.to(dev), c, uc)
    n_sched = num_sigmas - 1
    x_start = x.clone()

    def step(i, x):
        return sampler.step_call(denoiser, x, i % n_sched, s_in, sigmas, num_sigmas, cond, ucond)

    def barrier():
        if use_dist:
            torch.distributed.barrier()

    for i in range(max(warmup, 3)):           # >= 3: the third fused step captures the HIP graph
        x = step(i, x)
    # the timed steps are steps 0 .. K-1 of the schedule, from the prepared start state (the steps the parity fixture holds);
    # the states of the first three are kept by reference (step() returns a fresh tensor): nothing is added to the timed region
    x, kept = x_start, []
    barrier(); torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(steps):
        x_next = step(i, x)
        if i < 3:
            kept.append((x, x_next))
        x = x_next
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t_start
    if not torch.isfinite(x).all():
        raise SystemExit("non-finite latents after the timed steps")
    parity = check_timed_steps(parity_fx, fx_name, kept, sigmas) if parity_fx is not None else None
    steppers = list(unet.runtime(dev).steppers.values())
    graphed = bool(steppers) and steppers[0].graph is not None
    two_stream = bool(getattr(unet.runtime(dev), "last_forward_two_stream", False))
    # per-kernel breakdown: the same K steps again, launched eagerly with HIP events around every kernel
    prof = ops.Profiler() if profile else None
    if prof is not None:
        ops.PROFILER = prof
        t_p = time.perf_counter()
        for i in range(steps, 2 * steps):
            x = step(i, x)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t_p) / steps * 1e3
        ops.PROFILER = None
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    ms_per_step = elapsed / steps * 1e3
    value = world * steps / elapsed
    out = {
        "metric": "denoise-steps/sec (UNet fwd) at 16 views x 1024^2" if stage == 2 and T == 16
                  else f"denoise-steps/sec (UNet fwd) at {T} views x {lat * 8}^2",
        "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp8qk": "fp8-qk/bf16", "fp8": "fp8-attention/bf16"}[attn], "data": "synthetic",
        "config": {"workload": f"Hi3D stage-{stage} VideoUNet sampler step, {T} views @ {lat * 8}x{lat * 8} "
                               f"(CFG batch {2 * T}, latent {lat}x{lat}, in_channels {cfg['in_channels']}), "
                               "EulerEDM 25-step schedule, random-init 1.52B-param UNet",
                   "global_batch": 2 * T * world, "parallelism": f"replicas x{world} (one clip per GPU)",
                   "step_launch": "one HIP-graph replay per step" if graphed else "eager kernel launches",
                   "streams": "two HIP streams inside the step: the unconditional / conditional halves of the batch through the "
                              "two largest resolution levels side by side (HI3D_TWO_STREAM=auto)" if two_stream else "one"},
    }
    out["parity_checked"] = parity is not None
    if parity is not None:
        parity["launch_mode"] = ("HIP-graph replay" if graphed else "eager launches") + (", two HIP streams" if two_stream else "")
        out["parity"] = parity
    step_tf = STEP_TFLOP[stage] * (T / 16.0)
    summ = prof.summary() if prof is not None else None
    # executed work per step: what the kernels of this step actually compute (per-launch count of the profiled steps)
    exec_tf = (sum(d["flops"] for d in summ.values()) / steps / 1e12) if summ else STEP_TFLOP_EXECUTED[stage] * (T / 16.0)
    out["step_roofline"] = {"bound": "mfma", "achieved": round(exec_tf / (ms_per_step / 1e3), 1),
                            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(exec_tf / (ms_per_step / 1e3) / PEAK_BF16_TFLOPS, 4),
                            "executed_tflop_per_step": round(exec_tf, 2), "reference_graph_tflop_per_step": round(step_tf, 2),
                            "note": "whole step: EXECUTED TFLOP (reference graph minus the algebraically eliminated single-key "
                                    "cross-attention, SURVEY 8d) / wall time"}
    if prof is not None:
        total_ms = sum(d["ms"] for d in summ.values())
        fams = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
        for fam, d in fams:
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
            log(f"[bench] {fam:16s} {d['ms'] / steps:9.3f} ms/step  {d['launches'] // steps:4d} launches/step  "
                f"{tf:8.1f} TFLOP/s  {gbs:8.1f} GB/s(alg)")
        log(f"[bench] kernels total {total_ms / steps:.2f} ms/step; wall {ms_per_step:.2f} ms/step "
            f"({'graph replay' if graphed else 'eager'}), {eager_ms:.2f} ms/step eager with events")
        if a.shapes:
            for fam, d in sorted(prof.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"])[:90]:
                tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
                gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                log(f"[bench]   {fam:60s} {d['ms'] / steps:8.3f} ms/step {d['launches'] // steps:4d}x "
                    f"{d['ms'] / d['launches']:7.3f} ms each {tf:7.1f} TFLOP/s {gbs:7.0f} GB/s(alg)")
        # dominant kernel: the bf16 MFMA GEMM / implicit-GEMM conv kernel (gemm_bf16_kernel, all A-gather modes)
        g = [d for f, d in summ.items() if f.startswith("gemm_")]   # (the fused feed-forward kernel is listed on its own line)
        g_ms, g_fl, g_n = sum(d["ms"] for d in g), sum(d["flops"] for d in g), sum(d["launches"] for d in g)
        at = summ.get("attn_d64", summ.get("attn_d64_fp8qk", summ.get("attn_d64_fp8", dict(ms=0.0, flops=0.0, launches=1))))
        dom_is_gemm = g_ms >= at["ms"]
        k_ms, k_fl, k_n = (g_ms, g_fl, g_n) if dom_is_gemm else (at["ms"], at["flops"], at["launches"])
        k_by = sum(d["bytes"] for d in g) if dom_is_gemm else at.get("bytes", 0.0)
        ach = k_fl / (k_ms * 1e-3) / 1e12
        # HBM traffic per launch of that kernel from rocprofv3 PMC passes of this same command
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE; tools/pmc_traffic.py);
        # measured offline because counters serialise kernels -- null if no profile is committed
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{('s1' if stage == 1 else 's2')}.json")
        if os.path.exists(tpath) and T == 16 and attn == "bf16":      # the PMC passes were taken on this configuration
            traffic = json.load(open(tpath)).get("gemm_bf16_kernel" if dom_is_gemm else "attn_d64_kernel", {}).get("bytes_per_launch")
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_bf16_kernel" if dom_is_gemm else "attn_d64_kernel",
                           "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                           # algorithmic bytes per launch (every operand element once: ops.gemm's count), beside the measured traffic
                           "algorithmic_bytes_per_launch": int(k_by / k_n),
                           "traffic_over_algorithmic": round(traffic / (k_by / k_n), 3) if traffic and k_by else None,
                           "launches_per_step": k_n // steps, "avg_launch_ms": round(k_ms / k_n, 4),
                           "share_of_step": round(k_ms / steps / ms_per_step, 3),
                           "timing": f"HIP events around every launch over the {steps} steps after the timed region "
                                     "(the timed steps are graph replays)" +
                                     ("; in this pass the two CFG halves of the large levels run one after the other on ONE stream "
                                      "(same kernels and shapes as the timed region, which overlaps them on two streams): per-kernel "
                                      "durations free of the overlap, their sum exceeds ms_per_step by what the overlap hides"
                                      if two_stream else "")}
        out["kernels_ms_per_step"] = {f: round(d["ms"] / steps, 3) for f, d in fams}
        for fam in ("attn_d64", "attn_d64_fp8qk", "attn_d64_fp8"):
            if fam not in summ:
                continue
            d = summ[fam]
            a_tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            out["attention_mfma"] = {"kernel": fam, "achieved": round(a_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(a_tf / PEAK_BF16_TFLOPS, 4), "peak_is": "dense bf16 MFMA"}
            if fam != "attn_d64":
                # the fp8 kernels run their matrix products at the fp8 rate: state the fraction of THAT peak too (fp8qk: only
                # the score product -- half the FLOPs -- is fp8, so its ceiling is the harmonic mix 2 / (1/2.5 + 1/5) PF)
                pk = PEAK_FP8_TFLOPS if fam == "attn_d64_fp8" else 2.0 / (1.0 / PEAK_BF16_TFLOPS + 1.0 / PEAK_FP8_TFLOPS)
                out["attention_mfma"].update(frac_fp8_peak=round(a_tf / pk, 4), fp8_peak=round(pk, 1),
                                             fp8_peak_is="dense fp8 MFMA 5 PF" if fam == "attn_d64_fp8" else
                                             "score product at the 5 PF fp8 rate, P V at the 2.5 PF bf16 rate")
        for fam in ("groupnorm_silu", "layernorm"):
            if fam in summ:
                d = summ[fam]
                out[fam + "_hbm"] = {"achieved": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                     "unit": "GB/s", "frac": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        step_flops_exec = sum(d["flops"] for d in summ.values()) / steps
    else:
        step_flops_exec = None

    if cpu_sd is not None:
        port = cpu_baseline(cpu_sd, cfg, stage, unet, dev, step_flops_exec, step_tf)
        out["cpu_baseline"] = port
        if parity_fx is not None and parity_fx.get("ref_step_seconds"):
            # the reference's OWN classes on this exact configuration: their wall time per denoise step was
            # recorded when the parity fixture was generated (oracle/gen_golden.py, build container: the Python reference cannot
            # travel to the GPU box); the port timed live on THIS host's cores stays beside it
            ss, h = parity_fx["ref_step_seconds"], parity_fx["ref_host"]
            mean_s = sum(ss) / len(ss)
            out["cpu_baseline"] = {
                "value": round(1.0 / mean_s, 6), "unit": "steps/s", "cores": h["threads"], "kind": "reference",
                "seconds_per_step": [round(t, 1) for t in ss],
                "sample": f"{len(ss)} full denoise steps of THIS configuration (CFG batch {2 * T} at latent {lat}x{lat}, full width) through "
                          "the reference's EulerEDMSampler.step_call + LinearPredictionGuider + Denoiser + VideoUNet, fp32, no extrapolation; "
                          f"recorded by oracle/gen_golden.py in tests/golden/{fx_name}.pt on {h['where']}: {h['cpu']}, {h['threads']} torch "
                          f"threads on {h['cores']} cores, torch {h['torch']} (the Python reference does not travel to the GPU box)",
                "host": h, "port_on_this_host": port}
    return out, unet, sampler, ms_per_step


PARITY_TOL, PARITY_COS = 2.5e-2, 0.999


def check_timed_steps(fx, fx_name, kept, sigmas):
    """The first timed steps against the reference-class fixture (same bound as tests/test_timed_path_gpu.py): the guided
    denoised estimate D_i = (x_{i+1} - x_i s_{i+1}/s_i) / (1 - s_{i+1}/s_i) of every kept step vs the reference's
    EDMSampler.denoise output -- the state itself hides the network at sigma ~ 500 (|x| ~ 2800, |D| ~ 5).  Exits non-zero on a
    miss: a throughput number from a step that computes something else is not a measurement."""
    if not torch.allclose(sigmas.float().cpu(), fx["sigmas"], rtol=1e-6, atol=0):
        raise SystemExit("parity fixture: sigma schedule differs from the reference's")
    per_step = []
    for i, (x, x_next) in enumerate(kept[:fx["n_run"]]):
        r = (sigmas[i + 1] / sigmas[i]).double()
        D = ((x_next.double() - x.double() * r) / (1.0 - r)).float().cpu()
        ref = fx["denoised_f16"][i].float()
        rel = ((D - ref).abs().max() / ref.abs().max()).item()
        cs = torch.nn.functional.cosine_similarity(D.flatten(), ref.flatten(), dim=0).item()
        per_step.append({"step": i, "denoised_rel": round(rel, 5), "cos": round(cs, 6)})
        log(f"[bench] parity of timed step {i} vs the reference classes ({fx_name}.pt): denoised rel {rel:.4f} cos {cs:.6f}")
        if not (rel < PARITY_TOL and cs > PARITY_COS):
            raise SystemExit(f"PARITY FAILURE in the timed region: step {i} denoised rel {rel:.4f} cos {cs:.6f} "
                             f"(bounds {PARITY_TOL} / {PARITY_COS}) vs tests/golden/{fx_name}.pt")
    return {"fixture": f"tests/golden/{fx_name}.pt (the reference's EulerEDMSampler.step_call + guider + denoiser + VideoUNet, fp32 CPU)",
            "compared": "guided denoised estimate of each of the first timed steps, recovered from the two fp32 states around it",
            "steps": per_step, "tolerance": {"denoised_rel": PARITY_TOL, "cos": PARITY_COS}}


def clip_parallel_leg(a, unet, guider, stage, T, lat, dev, world, replica_ms, steps=None, cfg_split=None, overlap=False):
    """One clip of T views on all `world` GPUs: 2 (CFG halves) x world/2 (frame<->space all-to-all groups) when world is
    even, else 1 x world.  Every rank holds the same conditioning (seed 0) and the full latent; per step:
    76 all-to-alls + 44 GroupNorm sum all-reduces inside each half, one all-gather of the network output."""
    import torch.distributed as dist
    from hi3d_hip import synth
    from hi3d_hip.parallel import ClipParallelStepper
    from sgm.modules.diffusionmodules.discretizer import EDMDiscretization
    steps = a.steps if steps is None else steps
    if cfg_split is None:
        cfg_split = 2 if world % 2 == 0 else 1
    sp = world // cfg_split
    if T % sp or (lat // 8) ** 2 % sp:
        return {"skipped": f"frames ({T}) / lowest-level pixels ({(lat // 8) ** 2}) not divisible by sp={sp}"}
    stepper = ClipParallelStepper(unet, guider, T, cfg=cfg_split, overlap=overlap)
    comms = [stepper.comm] + ([stepper.comm2] if stepper.comm2 is not None else [])
    cnt = lambda: (sum(c.n_switches for c in comms), sum(c.n_allreduce for c in comms), sum(c.bytes_moved for c in comms), stepper.gather_bytes)
    x0, c, uc = synth.synth_conditioning(T, lat, lat, stage=stage, seed=0)
    c = {k: v.to(dev) for k, v in c.items()}
    uc = {k: v.to(dev) for k, v in uc.items()}
    sigmas = EDMDiscretization(sigma_max=700.0)(25, device=dev)
    x = (x0.to(dev) * torch.sqrt(1.0 + sigmas[0] ** 2.0)).contiguous()
    ioi = torch.zeros(2 // cfg_split, T, device=dev)
    n = len(sigmas) - 1
    for i in range(2):
        x = stepper.step(x, sigmas, i % n, c, uc, ioi)
    c0 = cnt()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + steps):
        x = stepper.step(x, sigmas, i % n, c, uc, ioi)
    torch.cuda.synchronize(); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if not torch.isfinite(x).all():
        raise RuntimeError("non-finite latents in the clip-parallel leg")
    ms = el.item() / steps * 1e3
    k = steps
    c1 = cnt()
    return {"mapping": f"cfg{cfg_split} x sp{sp}" + (" with overlap: the two CFG halves of a rank as two chains on two HIP streams, one "
                                                     "RCCL communicator each" if stepper.comm2 is not None else ""),
            "views": T, "steps": steps, "ms_per_step": round(ms, 2), "steps_per_s_one_clip": round(1e3 / ms, 4),
            "speedup_vs_one_gpu_replica": None if replica_ms is None else round(replica_ms / ms, 3), "scaling": "strong",
            "all_to_all_per_step": (c1[0] - c0[0]) // k,
            "gn_allreduce_per_step": (c1[1] - c0[1]) // k,
            "all_to_all_bytes_sent_per_rank_per_step": (c1[2] - c0[2]) // k,
            "all_gather_bytes_received_per_rank_per_step": (c1[3] - c0[3]) // k,
            "transport": "RCCL (torch.distributed nccl backend)"}


def vae_sharded_leg(a, T, lat, dev, world):
    """decode_first_stage of ONE clip of T frames sharded over all ranks + the RCCL all-gather that reassembles the clip on
    every rank (hi3d_hip.parallel.decode_sharded; reference hand-off: sgm/models/diffusion.py:117-135; north_star: "RCCL
    all-gather over xGMI only at VAE-decode hand-off").  Timed end to end (decode of T / world frames + gather), max over ranks."""
    import torch.distributed as dist
    from hi3d_hip import synth
    from hi3d_hip.parallel import decode_sharded
    from sgm.models.autoencoder import AutoencoderKL
    dd = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    synth.fill_module_(ae, 1, prefix="first_stage_model.")
    ae = ae.to(dev)
    z = torch.randn(T, 4, lat, lat, device=dev, generator=torch.Generator(device=dev).manual_seed(0))   # the same clip on every rank

    def dec(zz):
        return torch.cat([ae.decode(zz[i:i + 1]) for i in range(zz.shape[0])], 0)

    stats = {}
    out = decode_sharded(dec, z, stats=stats)                      # warm-up (weight re-layout, RCCL channel set-up)
    dist.barrier(); torch.cuda.synchronize()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        out = decode_sharded(dec, z, stats=stats)
    torch.cuda.synchronize(); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # the gather alone, on the decoded frames of the last call
    lo, hi = (T // world) * dist.get_rank(), (T // world) * (dist.get_rank() + 1)
    mine = out[lo:hi].contiguous() if T % world == 0 else None
    g_ms = None
    if mine is not None:
        buf = torch.empty_like(out)
        dist.all_gather_into_tensor(buf, mine)
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            dist.all_gather_into_tensor(buf, mine)
        torch.cuda.synchronize()
        g = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        g_ms = g.item() / 3 * 1e3
    if not torch.isfinite(out).all() or out.shape[0] != T:
        raise RuntimeError("bad output of the sharded decode")
    ms = el.item() / reps * 1e3
    gb = stats.get("gather_bytes", 0)
    return {"frames": T, "resolution": lat * 8, "frames_per_rank": T / world, "ms_per_clip": round(ms, 2),
            "frames_per_s_one_clip": round(T / ms * 1e3, 2), "all_gather_bytes_received_per_rank": gb,
            "all_gather_ms": None if g_ms is None else round(g_ms, 3),
            "all_gather_GBs_per_rank": None if not g_ms else round(gb / g_ms / 1e6, 1),
            "frame_dtype": str(out.dtype).replace("torch.", ""), "transport": "RCCL all_gather_into_tensor (torch.distributed nccl backend)"}


def cpu_baseline(cpu_sd, cfg, stage, unet, dev, step_flops_exec, step_tf):
    """Oracle (oracle/hi3d_oracle.py, fp32 torch CPU kernels = the reference's CPU path restated) timed on this box's host cores
    on a bounded sample of the same step: ONE full-width forward at T = 16, latent 32 x 32 (the clip length of the headline, 1/16
    of its pixels: ~4.4 executed TFLOP, 10 - 40 s on the box's cores) after a small warm-up call that touches the 6 GB of
    weights; extrapolated to the full step by executed FLOPs.  (Rounds 1-3 sampled T = 8 at latent 16 x 16 -- 1.1 TFLOP, too
    small to keep 128 threads busy: it under-stated the CPU about 3x per core against the reference-class probe of
    BASELINE.md section 3, which is quoted beside it.)"""
    from hi3d_hip import ops
    from oracle import hi3d_oracle as O
    T, hw = 16, 32
    g = torch.Generator().manual_seed(0)
    x = torch.randn((T, cfg["in_channels"], hw, hw), generator=g)
    ts = torch.full((T,), 0.25 * 1.5)
    ctx, y = torch.randn((1, 1, 1024), generator=g), torch.randn((1, cfg["adm_in_channels"]), generator=g)
    ioi = torch.zeros(1, T)
    # executed FLOPs of the sample, counted by the same per-launch formulae as the GPU path
    prof = ops.Profiler(); ops.PROFILER = prof
    unet(x.to(dev), ts.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=T, image_only_indicator=ioi.to(dev))
    torch.cuda.synchronize(); ops.PROFILER = None
    sample_flops = sum(d["flops"] for d in prof.summary().values())
    threads = torch.get_num_threads()
    times = []
    with torch.no_grad():
        # warm-up: first touch of the 6 GB of weights, thread pool start (T = 4, latent 8 x 8)
        O.video_unet(cpu_sd, cfg, x[:4, :, :8, :8].contiguous(), ts[:4], ctx, y, 4, ioi[:, :4], prefix="model.diffusion_model.")
        for _ in range(2):
            t0 = time.perf_counter()
            O.video_unet(cpu_sd, cfg, x, ts, ctx, y, T, ioi, prefix="model.diffusion_model.")
            times.append(time.perf_counter() - t0)
            if sum(times) > 25.0:
                break
    dt = min(times)
    cpu_tflops = sample_flops / dt / 1e12
    full = step_flops_exec if step_flops_exec else step_tf * 1e12
    return {"value": round(cpu_tflops * 1e12 / full, 6), "unit": "steps/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(), "tflops": round(cpu_tflops, 4),
            "extrapolation_factor": round(full / sample_flops, 2),      # full step / timed sample, by executed FLOPs
            "sample": f"oracle fp32 UNet forward, full width, T=16, latent 32x32 ({sample_flops / 1e12:.2f} TFLOP executed; best of "
                      f"{len(times)} run(s) after a small warm-up call = {dt:.1f}s = {cpu_tflops:.3f} TFLOP/s on {threads} torch "
                      f"threads, {os.cpu_count()} host CPUs), extrapolated to the {full / 1e12:.1f} TFLOP executed per full step",
            "reference_class_probe": "BASELINE.md section 3: the reference's own VideoUNet classes, full stage-1 size, 0.34 TFLOP/s on the "
                                     "8 cores of the survey container (not re-run here: /root/reference does not travel to the GPU box)",
            "seconds": round(dt, 2), "runs_s": [round(t, 2) for t in times]}


VAE_TFLOP_PER_FRAME = {512: 2.51, 1024: 10.47}     # decode_first_stage, BASELINE.md section 2


def bench_vae(a, rank, world, dev, use_dist, steps, warmup):
    """decode_first_stage of one clip: `--views` frames at 1024x1024 (latent 128x128) through the full-width
    AutoencoderKL decoder, en_and_decode_n_samples_a_time = 1 as configs/inference-v02.yaml ships.
    A "step" is the decode of the whole clip; value = frames/s."""
    from hi3d_hip import ops, synth
    from sgm.models.autoencoder import AutoencoderKL
    dd = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    synth.fill_module_(ae, 1, prefix="first_stage_model.")
    ae = ae.to(dev)
    T, lat = a.views, 128
    z = torch.randn(T, 4, lat, lat, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))

    from hi3d_hip import runtime_vae

    def clip():     # (the chunk loop of DiffusionEngine.decode_first_stage: HI3D_VAE_STREAMS=2 alternates frames over two streams)
        return runtime_vae.run_chunks(lambda lo, hi: ae.decode(z[lo:hi]), [(i, i + 1) for i in range(T)], dev)[-1]

    for _ in range(max(1, warmup)):
        out = clip()
    if use_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = clip()
    torch.cuda.synchronize()
    if use_dist:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(out).all()
    prof = ops.Profiler(); ops.PROFILER = prof
    clip(); torch.cuda.synchronize(); ops.PROFILER = None
    ms_frame = elapsed / steps / T * 1e3
    tf_ref = VAE_TFLOP_PER_FRAME[lat * 8]
    # EXECUTED work per frame, counted per launch by the profiler like the UNet lines (the up-sampling convs run as four 2x2 phase
    # convs: 8.92 of the reference graph's 10.47 TFLOP at 1024^2); the reference-graph fraction is reported beside it, labelled
    tf = sum(d["flops"] for d in prof.summary().values()) / T / 1e12
    res = {"metric": f"VAE decode frames/sec at {lat * 8}^2 (decode_first_stage)", "value": round(world * T * steps / elapsed, 3),
           "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 2),
           "ms_per_frame": round(ms_frame, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic",
           "config": {"workload": f"AutoencoderKL decoder (ch 128, mult 1-2-4-4), {T} frames @ {lat * 8}x{lat * 8}, one frame per call",
                      "streams": max(1, min(runtime_vae.VAE_STREAMS, T))},
           "roofline": {"bound": "mfma", "achieved": round(tf / (ms_frame / 1e3), 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / (ms_frame / 1e3) / PEAK_BF16_TFLOPS, 4), "traffic": None,
                        "executed_tflop_per_frame": round(tf, 3), "reference_graph_tflop_per_frame": tf_ref,
                        "frac_on_reference_graph_flops": round(tf_ref / (ms_frame / 1e3) / PEAK_BF16_TFLOPS, 4),
                        "note": "EXECUTED TFLOP per frame (per-launch count, as the UNet lines) / wall time per frame; "
                                "frac_on_reference_graph_flops prices the same time against the reference graph's work"}}
    summ = prof.summary(by_shape=True)
    for fam, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:30]:
        tfk = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        log(f"[bench]   {fam:64s} {d['ms'] / T:8.3f} ms/frame {d['launches'] // T:3d}x {tfk:7.1f} TFLOP/s {gbs:7.0f} GB/s(alg)")
    fams = prof.summary()
    res["kernels_ms_per_frame"] = {f: round(d["ms"] / T, 3) for f, d in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}
    return res


if __name__ == "__main__":
    main()
