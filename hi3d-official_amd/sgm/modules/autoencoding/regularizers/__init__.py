"""Config target only: the posterior sample/mode is fused into hi3d_vae_posterior
(reference: sgm/modules/autoencoding/regularizers/__init__.py:13-31)."""


class DiagonalGaussianRegularizer:
    def __init__(self, sample=True):
        self.sample = sample
