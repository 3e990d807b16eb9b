"""Debug aid (round 4): decode random latents through a ch=64 AutoencoderKL with HI3D_UP_PHASES / HI3D_VAE_STREAMS combinations and
report NaNs / differences per frame."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hi3d-official_amd"))
import torch
from hi3d_hip import synth, runtime_vae, ops
from sgm.models.autoencoder import AutoencoderKL
from sgm.models.diffusion import DiffusionEngine

dev = torch.device("cuda:0")
for ch in (64, 128):
    dd = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=ch,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    synth.fill_module_(ae, 6, prefix="first_stage_model.")
    ae = ae.to(dev)
    eng = DiffusionEngine.__new__(DiffusionEngine)
    torch.nn.Module.__init__(eng)
    eng.first_stage_model = ae
    eng.scale_factor, eng.en_and_decode_n_samples_a_time = 0.18215, 1
    for lat in (16, 32):
        z = torch.randn(16, 4, lat, lat, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        outs = {}
        for ph in (False, True):
            for ns in (1, 2):
                runtime_vae.UP_PHASES, runtime_vae.VAE_STREAMS = ph, ns
                for rep in range(2):
                    o = eng.decode_first_stage(z)
                    torch.cuda.synchronize()
                    nan = torch.isnan(o).flatten(1).any(1).nonzero().flatten().tolist()
                    outs[(ph, ns, rep)] = o
                    print(f"ch {ch} lat {lat} phases {ph} streams {ns} rep {rep}: NaN frames {nan}  absmax {o[~torch.isnan(o)].abs().max().item():.3f}")
        a, b = outs[(False, 1, 0)], outs[(True, 1, 0)]
        print(f"   phases vs nine-tap (1 stream): rel {((a - b).abs().max() / a.abs().max()).item():.3e}; 2 streams equal 1 stream (phases): "
              f"{torch.equal(outs[(True, 1, 0)], outs[(True, 2, 0)])} / {torch.equal(outs[(True, 1, 0)], outs[(True, 2, 1)])}")
