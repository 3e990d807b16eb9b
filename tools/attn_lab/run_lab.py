"""Round-3 incident lab, GPU side: for every variant library built by make_variants.py, launch the d=64 flash attention
many times on fixed inputs at ragged sequence lengths and count launches whose output differs from the first; for the
failing launches, work out WHICH quantity of WHICH lanes went wrong:

  * rows (-> wave, query block, lane group), heads, batches
  * fit  bad_row = a * good_row + c * v_last  (v_last = V of the only valid key of the ragged tile at S % 64 == 1):
      a != 1, c == 0, no residual   -> the row sum l is wrong (P of masked keys entered it)
      residual confined to some d   -> those O accumulator registers are wrong
  * fp32 reference of the same rows (which of the two launches is the wrong one)

usage: run_lab.py [launches] [variant ...]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAB = os.path.join(ROOT, "gpurun_tmp", "attn_lab")
dev = torch.device("cuda:0")


def load(name):
    lib = C.CDLL(os.path.join(LAB, name, "lib.so"))
    i32, f32, vp = C.c_int32, C.c_float, C.c_void_p
    lib.hi3d_attn_d64.restype = C.c_int
    lib.hi3d_attn_d64.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.hi3d_transpose_v.restype = C.c_int
    lib.hi3d_transpose_v.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.hi3d_last_error.restype = C.c_char_p
    return lib


def reference_rows(qkv, B, S, H, b, h, rows, scale):
    Cc = H * 64
    x = qkv[b * S:(b + 1) * S].float()
    q = x[rows, h * 64:(h + 1) * 64]
    k = x[:, Cc + h * 64:Cc + (h + 1) * 64]
    v = x[:, 2 * Cc + h * 64:2 * Cc + (h + 1) * 64]
    p = torch.softmax(q @ k.t() * scale, dim=-1)
    return p @ v, v[S - 1]


def forensics(qkv, B, S, H, good, bad, scale):
    Cc = H * 64
    d = (bad.float() - good.float()).abs()
    rows = d.amax(dim=1).nonzero().flatten().tolist()
    grp = {}
    for r in rows:
        b, tok = divmod(r, S)
        heads = d[r].view(H, 64).amax(dim=1).nonzero().flatten().tolist()
        for h in heads:
            grp.setdefault((b, h), []).append(tok)
    for (b, h), toks in list(grp.items())[:3]:
        toks = sorted(toks)
        w_rows = [t % 64 for t in toks]
        print(f"    batch {b} head {h}: {len(toks)} rows, tokens {toks[0]}..{toks[-1]} (row within wave {w_rows[0]}..{w_rows[-1]}, "
              f"q tile {toks[0] // 256}, wave {(toks[0] % 256) // 64})")
        ref, v_last = reference_rows(qkv, B, S, H, b, h, toks, scale)
        g = good[[b * S + t for t in toks], h * 64:(h + 1) * 64].float()
        x = bad[[b * S + t for t in toks], h * 64:(h + 1) * 64].float()
        print(f"      |good - fp32 ref| max {float((g - ref).abs().max()):.3e}   |bad - fp32 ref| max {float((x - ref).abs().max()):.3e}")
        for i in (0, len(toks) // 2, len(toks) - 1):
            A = torch.stack([g[i], v_last], dim=1)                       # [64, 2]
            sol = torch.linalg.lstsq(A, x[i].unsqueeze(1)).solution.flatten()
            res = x[i] - A @ sol
            blk = res.view(8, 8).abs().amax(dim=1)
            print(f"      token {toks[i]}: bad = {float(sol[0]):.4f} * good + {float(sol[1]):.4f} * v_last, residual max {float(res.abs().max()):.3e} "
                  f"(|bad-good| max {float((x[i] - g[i]).abs().max()):.3e}); residual per 8-d block: " + " ".join(f"{float(t):.1e}" for t in blk))
            nz = (x[i] - g[i]).abs() > 0
            print(f"        differing d columns: {nz.nonzero().flatten().tolist()}")


def show_dump(first, bad, B, S, H, names=("m_before", "own_max", "tile_max", "l_before", "psum", "l_after")):
    """n_dump variant: columns H*64.. of every output row hold, per (head, half-wave), six floats of the peeled tile's softmax
    state: m before, own max, tile max, l before, row sum, l after."""
    Cc = H * 64
    d = (bad[:, :Cc].float() - first[:, :Cc].float()).abs()
    rows = d.amax(dim=1).nonzero().flatten().tolist()
    shown = 0
    for r in rows:
        heads = d[r].view(H, 64).amax(dim=1).nonzero().flatten().tolist()
        for h in heads:
            if shown >= 6:
                return
            if shown % 3 != 0 and (r % 8):
                continue
            shown += 1
            for tag, t in (("first", first), ("bad  ", bad)):
                v = t[r, Cc:].contiguous().view(torch.float32).view(H, 2, 6)[h].cpu()
                print(f"      row {r} (token {r % S}, batch {r // S}) head {h} {tag}: " +
                      " | ".join("half%d " % hf + " ".join(f"{names[i]}={v[hf, i].item():.6g}" for i in range(6)) for hf in (0, 1)))
    # how many (row, head, half, field) entries differ at all, by field
    a = first[:, Cc:].contiguous().view(torch.float32).view(-1, H, 2, 6)
    b = bad[:, Cc:].contiguous().view(torch.float32).view(-1, H, 2, 6)
    ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
    print("      differing dump entries by field: " + ", ".join(f"{names[i]}={int(ne[..., i].sum())}" for i in range(6)))


def run(name, launches, B, H, S, scale):
    lib = load(name)
    Cc = H * 64
    dump = name.endswith("dump")
    ldo = Cc + (24 * H if dump else 0)
    g = torch.Generator().manual_seed(1)
    S_pad = (S + 63) // 64 * 64
    # 64 rows of padding behind the last batch: the `nooob` variant reads K rows beyond S (masked afterwards)
    qkv_all = (torch.randn((B * S + 64, 3 * Cc), generator=g) * 1.5).to(torch.bfloat16).to(dev)
    qkv = qkv_all[:B * S]
    vt = torch.empty((B, H, 64, S_pad), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.hi3d_transpose_v(qkv[:, 2 * Cc:].data_ptr(), vt.data_ptr(), B, H, S, S_pad, 3 * Cc, st)
    assert rc == 0, lib.hi3d_last_error()
    RING = 40
    ring = [torch.zeros((B * S, ldo), device=dev, dtype=torch.bfloat16) for _ in range(RING)]
    first = torch.zeros((B * S, ldo), device=dev, dtype=torch.bfloat16)

    def launch(o):
        rc = lib.hi3d_attn_d64(qkv.data_ptr(), qkv[:, Cc:].data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, S, S, 3 * Cc, 3 * Cc, S_pad, ldo,
                               scale, st)
        assert rc == 0, lib.hi3d_last_error()
    launch(first)
    torch.cuda.synchronize()
    nbad, shown, rowsets = 0, 0, {}
    done = 0
    while done < launches:
        n = min(RING, launches - done)
        for i in range(n):
            launch(ring[i])
        flags = torch.stack([(ring[i][:, :Cc] != first[:, :Cc]).any() for i in range(n)]).tolist()
        for i in range(n):
            if flags[i]:
                nbad += 1
                d = (ring[i][:, :Cc].float() - first[:, :Cc].float()).abs().amax(dim=1)
                rws = d.nonzero().flatten()
                key = tuple(sorted(set((rws % S % 64).tolist())))
                rowsets[key] = rowsets.get(key, 0) + 1
                if shown < 2:
                    shown += 1
                    print(f"  launch {done + i}: {rws.numel()} rows differ, max |diff| {float(d.max()):.3e}")
                    if dump:
                        show_dump(first, ring[i], B, S, H, *({"n_dump": (), "n3_dump": (("l_tile-3", "l_tile-2", "l_tile-1", "l_peel_entry", "psum_tile-1", "psum_peel"),)}.get(
                                      name, (("l_entry", "d", "alpha", "l_scaled", "m_after", "psum"),))))
                    else:
                        forensics(qkv, B, S, H, first[:, :Cc], ring[i][:, :Cc], scale if scale else 1.0 / 1.4426950408889634)
        done += n
    print(f"{name:9s} B={B} H={H} S={S} scale={scale:g}: {nbad} of {launches} launches differ from the first"
          + ("; rows-within-wave patterns: " + "; ".join(f"{list(k)[:2]}..{list(k)[-1:]} x{v}" for k, v in rowsets.items()) if rowsets else ""))
    return nbad


def timing(names, B=32, H=5, S=16384, reps=8):
    """ms per launch at the UNet's largest attention (pre-scaled q), variants interleaved on the same box."""
    Cc = H * 64
    g = torch.Generator().manual_seed(2)
    qkv = (torch.randn((B * S, 3 * Cc), generator=g) * 0.5).to(torch.bfloat16).to(dev)
    vt = torch.empty((B, H, 64, S), device=dev, dtype=torch.bfloat16)
    out = torch.empty((B * S, Cc), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    libs = {n: load(n) for n in names}
    libs[names[0]].hi3d_transpose_v(qkv[:, 2 * Cc:].data_ptr(), vt.data_ptr(), B, H, S, S, 3 * Cc, st)
    res = {n: [] for n in names}
    outs = {}
    for rep in range(reps + 1):
        for n in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                libs[n].hi3d_attn_d64(qkv.data_ptr(), qkv[:, Cc:].data_ptr(), vt.data_ptr(), out.data_ptr(), B, H, S, S, 3 * Cc, 3 * Cc, S, Cc,
                                      0.0, st)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                res[n].append(e0.elapsed_time(e1) / 3)
            else:
                outs[n] = out.clone()
    fl = 4.0 * B * H * S * S * 64
    for n in names:
        ms = sorted(res[n])[len(res[n]) // 2]
        same = torch.equal(outs[n], outs[names[0]])
        print(f"timing {n:9s} B={B} H={H} S={S}: median {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s), min {min(res[n]):.3f}; output == {names[0]}: {same}")


def main():
    args = sys.argv[1:]
    launches = int(args[0]) if args and args[0].isdigit() else 1000
    names = [a for a in args if not a.isdigit()] or sorted(os.listdir(LAB))
    print(torch.cuda.get_device_name(0))
    for name in names:
        for (B, H, S) in ((16, 12, 577), (16, 12, 513)) + (((16, 12, 576),) if name == "n3_dump" else ()):
            run(name, launches, B, H, S, 0.125)
    tn = [n for n in ("r2ship", "unified") if n in names]
    if len(tn) == 2:
        timing(tn)
        timing(tn, B=32, H=10, S=4096)
    # the pre-scaled form (scale == 0: q carries scale * log2 e) of the kernels that matter
    for name in [n for n in names if n in ("old", "unified")]:
        run(name, launches, 16, 12, 577, 0.0)


if __name__ == "__main__":
    main()
