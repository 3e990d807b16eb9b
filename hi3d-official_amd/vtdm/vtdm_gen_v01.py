"""Stage-1 VideoLDM (reference: vtdm/vtdm_gen_v01.py:24-76): DeepSpeed/.ckpt/.safetensors
loader and `add_custom_cond` (cond_aug noise on the conditioning frame)."""
import torch

from sgm.models.diffusion import DiffusionEngine


class VideoLDM(DiffusionEngine):
    def __init__(self, num_samples, trained_param_keys=("",), *args, **kwargs):
        self.trained_param_keys = list(trained_param_keys)
        super().__init__(*args, **kwargs)
        self.num_samples = num_samples

    def init_from_ckpt(self, path):
        if path.endswith("ckpt"):
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd)
        elif path.endswith("pt"):               # DeepSpeed ZeRO dump: {'module': {'module.<key>': tensor}}
            raw = torch.load(path, map_location="cpu")["module"]
            sd = {k[len("module."):]: v for k, v in raw.items()}
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            raise NotImplementedError(path)
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if missing:
            print(f"Missing Keys: {missing}")
        if unexpected:
            print(f"Unexpected Keys: {unexpected}")

    @torch.no_grad()
    def add_custom_cond(self, batch, infer=False):
        batch["num_video_frames"] = self.num_samples
        image = batch["video"][:, :, 0]
        n, dev = image.shape[0], image.device
        batch["cond_frames_without_noise"] = image.half()
        if infer:
            cond_aug = torch.full((n,), 0.02, device=dev)
        else:
            cond_aug = torch.exp(-3.0 + 0.5 * torch.randn((n,), device=dev))
        batch["cond_aug"] = cond_aug.half()
        batch["cond_frames"] = (image + cond_aug.reshape(n, 1, 1, 1) * torch.randn_like(image)).half()
        if "image_only_indicator" not in batch:
            batch["image_only_indicator"] = torch.zeros((n, self.num_samples), device=dev).half()
        return batch
