import numpy as np


def label_encoder(adata, encoder, condition_key=None):
    """Integer-encode `adata.obs[condition_key]` with `encoder` (label -> int).

    Same contract as scarches/dataset/trvae/_utils.py:4-36: float64 array of length n_obs; labels missing from the
    encoder are reported and mapped to -1.
    """
    values = np.asarray(adata.obs[condition_key])
    labels = np.zeros(adata.shape[0])
    present = set(np.unique(values).tolist())
    unknown = present.difference(encoder.keys())
    if unknown:
        print(f"Warning: Labels in adata.obs[{condition_key}] is not a subset of label-encoder!")
        print(f"The missing labels are: {unknown}")
        print("Therefore integer value of those labels is set to -1")
        for lab in unknown:
            labels[values == lab] = -1
    for lab, code in encoder.items():
        labels[values == lab] = code
    return labels
