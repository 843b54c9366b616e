"""scarches_amd - MI355X-native conditional-VAE train/map step with the scArches API for that path.

    import scarches_amd as sca
    model = sca.models.TRVAE(adata, condition_key="study", ...)
    model.train(n_epochs=..., ...)
    q = sca.models.TRVAE.load_query_data(query_adata, model)
    q.get_latent()

All arithmetic runs in hand-written HIP kernels (scarches_amd/csrc, C ABI in include/scarches_hip.h); there is no CPU
fallback: importing works anywhere, using a model needs a gfx950 device and the built libscarches_hip.so.
"""
__version__ = "0.1.0"

from . import dataset, models, trainers, utils  # noqa: E402,F401
from .anndata_lite import CellData  # noqa: E402,F401
