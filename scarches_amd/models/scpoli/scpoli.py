"""`scpoli`: the network object behind `scPoli.model`, API of scarches/models/scpoli/scpoli.py:10-360.

Same constructor, attributes (`condition_encoders`, `conditions_combined_encoder`, `cell_type_encoder`,
`prototypes_labeled`, `prototypes_unlabeled`, `new_prototypes`, ...), parameter names (`embeddings.{i}.weight`,
`encoder.FC.L0.cond_L.weight` with `[H, sum(E)]`, `theta [G, n_conditions_combined]`) and
`forward(x, batch, combined_batch, sizefactor, celltypes, labeled) -> (z1, recon_loss, kl_div, mmd_loss)`;
the arithmetic runs in the HIP engine (arch="scpoli").
"""
import torch
import torch.nn as nn

from ..base._engine_module import EngineBackedModule
from ..trvae.modules import Decoder, Encoder


class scpoli(EngineBackedModule):
    def __init__(self, input_dim, hidden_layer_sizes, cell_types, unknown_ct_names, conditions, conditions_combined,
                 inject_condition, latent_dim, embedding_dims, embedding_max_norm, recon_loss, dr_rate, beta, use_bn,
                 use_ln, prototypes_labeled, prototypes_unlabeled):
        super().__init__()
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.embedding_dims = embedding_dims
        self.embedding_max_norm = embedding_max_norm
        self.cell_types = cell_types
        self.n_cell_types = len(cell_types)
        self.cell_type_encoder = {k: v for k, v in zip(cell_types, range(len(cell_types)))}
        self.n_conditions = [len(conditions[cond]) for cond in conditions.keys()]
        self.n_reference_conditions = None
        self.conditions = conditions
        self.condition_encoders = {cond: {k: v for k, v in zip(conditions[cond], range(len(conditions[cond])))}
                                   for cond in conditions.keys()}
        self.conditions_combined = conditions_combined
        self.n_conditions_combined = len(conditions_combined)
        self.conditions_combined_encoder = {k: v for k, v in zip(conditions_combined, range(len(conditions_combined)))}
        self.inject_condition = inject_condition
        self.use_bn = use_bn
        self.use_ln = use_ln
        self.use_mmd = False
        self.recon_loss = recon_loss
        self.hidden_layer_sizes = hidden_layer_sizes
        self.freeze = False
        self.unknown_ct_names = unknown_ct_names
        if self.unknown_ct_names is not None:
            for unknown_ct in self.unknown_ct_names:
                self.cell_type_encoder[unknown_ct] = -1
        self.prototypes_labeled = {"mean": None, "cov": None} if prototypes_labeled is None else prototypes_labeled
        self.prototypes_unlabeled = {"mean": None} if prototypes_unlabeled is None else prototypes_unlabeled
        self.new_prototypes = None
        self.num_reference_conditions = None
        if self.prototypes_labeled["mean"] is not None:
            # indices of cell types that have no prototype yet (filled in by the trainer)
            n_old = len(self.prototypes_labeled["mean"])
            self.new_prototypes = [n_old + i for i in range(self.n_cell_types - n_old)]
        self.dr_rate = dr_rate
        self.use_dr = self.dr_rate > 0
        if recon_loss not in ("nb", "zinb", "mse", "poisson"):
            raise NotImplementedError(f"recon_loss={recon_loss!r} is not built in the HIP path (nb, zinb, mse, poisson are); "
                                      "the reference's 'nb_dist' and 'bernoulli' branches are broken upstream")
        if recon_loss in ["nb", "zinb"]:
            self.theta = torch.nn.Parameter(torch.randn(self.input_dim, self.n_conditions_combined))
        else:
            self.theta = None

        enc_sizes = [self.input_dim] + list(self.hidden_layer_sizes)
        dec_sizes = list(reversed(self.hidden_layer_sizes)) + [self.input_dim]
        self.embeddings = nn.ModuleList(nn.Embedding(self.n_conditions[i], self.embedding_dims[i],
                                                     max_norm=self.embedding_max_norm)
                                        for i in range(len(self.embedding_dims)))
        print("Embedding dictionary:\n", f"\tNum conditions: {self.n_conditions}\n",
              f"\tEmbedding dim: {self.embedding_dims}")
        e_tot = sum(self.embedding_dims)
        self.encoder = Encoder(enc_sizes, self.latent_dim, self.use_bn, self.use_ln, self.use_dr, self.dr_rate,
                               e_tot if "encoder" in self.inject_condition else None)
        self.decoder = Decoder(dec_sizes, self.latent_dim, self.recon_loss, self.use_bn, self.use_ln, self.use_dr,
                               self.dr_rate, e_tot if "decoder" in self.inject_condition else None)
        self._engine = None
        self.seed = 2020

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _spec(self):
        return dict(arch="scpoli", G=self.input_dim, C=self.n_conditions_combined, H=list(self.hidden_layer_sizes),
                    Z=self.latent_dim, recon=self.recon_loss, use_bn=self.use_bn, use_ln=self.use_ln,
                    dr_rate=self.dr_rate, use_mmd=False, seed=self.seed, E=list(self.embedding_dims),
                    n_conds=list(self.n_conditions), inject=list(self.inject_condition),
                    max_norm=self.embedding_max_norm)

    # ------------------------------------------------------------------------------------------ reference API
    def forward(self, x=None, batch=None, combined_batch=None, sizefactor=None, celltypes=None, labeled=None):
        """(z1, recon_loss, kl_div, mmd_loss=0) for a dense minibatch (no autograd graph; the trainer's step runs
        forward + backward + Adam in the engine)."""
        eng, x, B = self._bind_dense(x)
        covs = batch.to(eng.device).to(torch.int32).contiguous()
        comb = combined_batch.to(eng.device).to(torch.int32).contiguous()
        eng.forward_backward(None, comb, B, train=self.training, compute_grad=False, accumulate_logs=False, covs=covs)
        L = eng.losses.clone()
        return eng.ws["z"][:B].clone(), L[0], L[1], torch.zeros((), device=eng.device)

    def get_latent(self, x, c=None, mean=False):
        """Latent sample (or mean) of a dense block `x` with covariate codes `c` [n, n_covariates]."""
        eng = self.engine(x.device if x.is_cuda else None)
        x = x.to(eng.device, torch.float32).contiguous()
        covs = None if c is None else c.to(eng.device).to(torch.int32).contiguous()
        z, mu, var = eng.encode(None, None, x.shape[0], mean=False, x_dense=x, covs=covs)
        return mu.clone() if mean else z.clone()

    def _proto_distances(self, latent, classes_list, get_prob=False):
        """(classes_list[winner], winner's distance / probability, [n, len(classes_list)] distances or softmax(-d)) of
        the latent rows against the labeled prototypes of one hierarchy level: sca_proto_classify (scpoli_ops.hip)."""
        from ... import _lib
        lib = _lib.load()
        dev = latent.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("scpoli.classify runs on a HIP device only (no CPU fallback)")
        protos = self.prototypes_labeled["mean"] = self.prototypes_labeled["mean"].to(dev)
        P = protos.to(torch.float32).contiguous()
        cl = classes_list.to(dev).to(torch.int32).contiguous()
        K = P.shape[0]
        if cl.numel() == 0 or int(cl.min()) < 0 or int(cl.max()) >= K:
            raise IndexError("classes_list holds a prototype index outside [0, %d)" % K)
        lat = latent.to(torch.float32).contiguous()
        n, Z = lat.shape
        dists = torch.empty(n, cl.numel(), dtype=torch.float32, device=dev)
        pred = torch.empty(n, dtype=torch.int32, device=dev)
        uncert = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(lib.sca_proto_classify(lat.data_ptr(), lat.stride(0) if n else Z, n, Z, P.data_ptr(), K,
                                          cl.data_ptr(), cl.numel(), int(bool(get_prob)), dists.data_ptr(),
                                          pred.data_ptr(), uncert.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "proto_classify")
        return pred.to(classes_list.dtype), uncert, dists

    def classify(self, x, c=None, prototype=False, classes_list=None, p=2, get_prob=False, log_distance=True):
        """Nearest labeled prototype (restricted to `classes_list`): returns (preds, uncertainty, distances)."""
        if p != 2:
            raise NotImplementedError("classify: only the euclidean distance (p = 2) is built in the HIP path")
        latent = x if prototype else self.get_latent(x, c)
        dev = latent.device
        if dev.type != "cuda":
            latent = latent.to(next(self.parameters()).device)
        return self._proto_distances(latent, classes_list.to(latent.device), get_prob)

    def add_new_cell_type(self, latent, cell_type_name, prototypes, classes_list=None):
        """Register a new cell type whose prototype is the mean of the given unlabeled prototypes; its covariance is
        estimated from the latent rows that fall closest to it."""
        from ...trainers.scpoli._utils import cov
        dev = next(self.parameters()).device
        self.cell_types.append(cell_type_name)
        self.n_cell_types += 1
        self.cell_type_encoder = {k: v for k, v in zip(self.cell_types, range(len(self.cell_types)))}
        classes_list = torch.cat((classes_list, torch.tensor([self.n_cell_types - 1], device=classes_list.device)))
        new_proto = self.prototypes_unlabeled["mean"][prototypes].mean(0).unsqueeze(0)
        self.prototypes_labeled["mean"] = torch.cat((self.prototypes_labeled["mean"], new_proto), dim=0).to(dev)
        latent = latent.to(dev)
        y_hat, _, _ = self._proto_distances(latent, classes_list.to(dev))
        rows = y_hat.eq(self.n_cell_types - 1).nonzero(as_tuple=False)[:, 0]
        new_cov = cov(latent[rows, :]).unsqueeze(0).to(self.prototypes_labeled["cov"].device)
        self.prototypes_labeled["cov"] = torch.cat((self.prototypes_labeled["cov"], new_cov), dim=0)
