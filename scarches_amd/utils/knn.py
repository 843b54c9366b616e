"""Weighted-kNN label transfer in the latent space on the GPU - API mirror of scarches/utils/knn.py:10-139
(`weighted_knn_trainer`, `weighted_knn_transfer`: same names, arguments, return values and error messages).

The reference fits an sklearn `KNeighborsTransformer` (brute force, euclidean) and then walks a Python double loop over
queries x label columns; here the reference embedding stays in HBM, `sca_knn_topk` finds the k nearest reference cells
of every query and `sca_knn_vote` evaluates the Gaussian-weighted vote of one label column for all queries at once."""
import numpy as np
import pandas as pd
import torch

from .. import _lib


class HipKNeighbors:
    """What `weighted_knn_trainer` returns: the reference embedding resident on the device (the counterpart of the
    fitted `KNeighborsTransformer`, knn.py:28-41)."""

    def __init__(self, train_emb, n_neighbors=50, device=None):
        self.lib = _lib.load()
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise _lib.HipLibraryError("scarches_amd runs on a HIP device only (no CPU fallback)")
        emb = np.asarray(train_emb.todense() if hasattr(train_emb, "todense") else train_emb, dtype=np.float32)
        if emb.ndim != 2:
            raise ValueError("the embedding must be a 2-D array")
        self.n_neighbors = int(n_neighbors)
        if self.n_neighbors > self.lib.sca_knn_max_k():
            raise ValueError(f"n_neighbors must be <= {self.lib.sca_knn_max_k()}")
        if self.n_neighbors > emb.shape[0]:
            raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {self.n_neighbors}, "
                             f"n_samples_fit = {emb.shape[0]}, n_samples = {emb.shape[0]}")
        if emb.shape[1] > 64:
            raise ValueError("embeddings wider than 64 features are not supported by the kNN kernel")
        self.ref = torch.from_numpy(np.ascontiguousarray(emb)).to(dev)
        self.device = dev

    def kneighbors_device(self, X):
        """(dist [Nq, k] float32 ascending, idx [Nq, k] int32) as device tensors."""
        q = X if torch.is_tensor(X) else torch.from_numpy(np.ascontiguousarray(
            np.asarray(X.todense() if hasattr(X, "todense") else X, dtype=np.float32)))
        q = q.to(self.device, torch.float32).contiguous()
        nr, d = self.ref.shape
        if q.ndim != 2 or q.shape[1] != d:
            raise ValueError(f"X has {q.shape[-1]} features, but the kNN model is expecting {d} features as input.")
        nq, k = q.shape[0], self.n_neighbors
        chunks = self.lib.sca_knn_num_chunks(nr, nq)
        pd2 = torch.empty(max(1, chunks * nq * k), dtype=torch.float32, device=self.device)
        pidx = torch.empty(max(1, chunks * nq * k), dtype=torch.int32, device=self.device)
        dist = torch.empty(nq, k, dtype=torch.float32, device=self.device)
        idx = torch.empty(nq, k, dtype=torch.int32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.sca_knn_topk(self.ref.data_ptr(), nr, q.data_ptr(), nq, d, k, pd2.data_ptr(),
                                         pidx.data_ptr(), dist.data_ptr(), idx.data_ptr(), st), "knn_topk")
        return dist, idx

    def kneighbors(self, X):
        dist, idx = self.kneighbors_device(X)
        return dist.cpu().numpy().astype(np.float64), idx.cpu().numpy().astype(np.int64)


def _embedding(adata, key, what):
    if key == "X":
        return adata.X
    if key in adata.obsm.keys():
        return adata.obsm[key]
    raise ValueError(f"{what} should be set to either 'X' or the name of the obsm layer to be used!")


def weighted_knn_trainer(train_adata, train_adata_emb, n_neighbors=50):
    """scarches/utils/knn.py:10-41."""
    print(f"Weighted KNN with n_neighbors = {n_neighbors} ... ", end="")
    return HipKNeighbors(_embedding(train_adata, train_adata_emb, "train_adata_emb"), n_neighbors)


def weighted_knn_transfer(query_adata, query_adata_emb, ref_adata_obs, label_keys, knn_model, threshold=1,
                          pred_unknown=False, mode="package"):
    """scarches/utils/knn.py:43-139: returns (pred_labels, uncertainties) DataFrames indexed by the query cells, one
    column per reference obs column whose name starts with `label_keys`."""
    if not type(knn_model) == HipKNeighbors:
        raise ValueError("knn_model should be of type scarches_amd.utils.knn.HipKNeighbors "
                         "(returned by weighted_knn_trainer)!")
    query_emb = _embedding(query_adata, query_adata_emb, "query_adata_emb")
    if mode != "package":
        raise Exception("Inquery Mode!")
    dist, idx = knn_model.kneighbors_device(query_emb)
    nq, k = dist.shape
    lib, dev = knn_model.lib, knn_model.device
    cols = ref_adata_obs.columns[ref_adata_obs.columns.str.startswith(label_keys)]
    uncertainties = pd.DataFrame(columns=cols, index=query_adata.obs_names)
    pred_labels = pd.DataFrame(columns=cols, index=query_adata.obs_names)
    st = torch.cuda.current_stream(dev).cuda_stream
    for j in cols:
        # codes in sorted label order: the reference walks np.unique(...) with a strict '<' (first maximum wins)
        uniq, codes = np.unique(ref_adata_obs[j].values, return_inverse=True)
        lab = torch.from_numpy(codes.astype(np.int32)).to(dev)
        pred = torch.empty(nq, dtype=torch.int32, device=dev)
        prob = torch.empty(nq, dtype=torch.float64, device=dev)
        _lib.check(lib.sca_knn_vote(dist.data_ptr(), idx.data_ptr(), nq, k, lab.data_ptr(), pred.data_ptr(),
                                    prob.data_ptr(), st), "knn_vote")
        best_prob = prob.cpu().numpy()
        best = uniq[pred.cpu().numpy()].astype(object)
        if pred_unknown:
            best[~(best_prob >= threshold)] = "Unknown"
        pred_labels[j] = best
        uncertainties[j] = np.maximum(1 - best_prob, 0)
    print("finished!")
    return pred_labels, uncertainties
