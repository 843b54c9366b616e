"""expiMap trainer: trVAETrainer + proximal operators after every Adam step.

API of scarches/trainers/expimap/regularized.py:5-361 (same constructor arguments, `use_prox_ops`, `corr_coeffs`,
`watch_lr`, annealing schedule and operator order).  `ProxGroupLasso` / `ProxL1` are callable objects like the
reference's, but act on device weights through the HIP kernels `sca_prox_group_lasso` / `sca_prox_l1`.
"""
import torch

from ... import _lib
from ..trvae.unsupervised import trVAETrainer


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class ProxGroupLasso:
    """Column-wise (dim-0 norm) group soft-threshold: W[:, c] *= max(0, 1 - coeff_c / ||W[:, c]||), coeff = alpha
    (* omega[c]).  In place on a contiguous [rows, cols] HIP tensor."""

    def __init__(self, alpha, omega=None, inplace=True):
        self._alpha = alpha
        self._omega = None if omega is None else omega.detach().reshape(-1).to(torch.float32)
        self._inplace = inplace

    def __call__(self, W):
        if not self._inplace:
            W = W.clone()
        assert W.is_cuda and W.is_contiguous() and W.dim() == 2
        om = None
        if self._omega is not None:
            om = self._omega = self._omega.to(W.device)
            assert om.numel() == W.shape[1]
        # column sums of squares in a workspace this operator owns (sca_prox_group_lasso keeps them in a file-scope
        # array of the library: two operators on different streams would race on it)
        ws = getattr(self, "_colsq", None)
        if ws is None or ws.numel() < W.shape[1] or ws.device != W.device:
            ws = self._colsq = torch.empty(W.shape[1], dtype=torch.float32, device=W.device)
        _lib.check(_lib.load().sca_prox_group_lasso_ws(W.data_ptr(), W.shape[0], W.shape[1], float(self._alpha),
                                                       _lib.ptr(om), ws.data_ptr(), _stream(W)), "prox_group_lasso")
        return W


class ProxL1:
    """Soft-threshold by alpha; with a mask `I` (weight layout) only entries where I == 0 are touched."""

    def __init__(self, alpha, I=None, inplace=True):
        self._I = None if I is None else I.detach().to(torch.float32).contiguous()
        self._alpha = alpha
        self._inplace = inplace

    def __call__(self, W):
        if not self._inplace:
            W = W.clone()
        assert W.is_cuda and W.is_contiguous()
        I = None
        if self._I is not None:
            I = self._I = self._I.to(W.device)
            assert I.shape == W.shape
        _lib.check(_lib.load().sca_prox_l1(W.data_ptr(), W.numel(), float(self._alpha), _lib.ptr(I), _stream(W)),
                   "prox_l1")
        return W


class expiMapTrainer(trVAETrainer):
    """Extra arguments over trVAETrainer: alpha (group lasso), omega (per-term coefficients), alpha_l1 (+ epoch
    annealing) for soft masks, gamma_ext (+ annealing) for unconstrained extension terms, beta (HSIC weight),
    print_stats."""

    def __init__(self, model, adata, alpha, omega=None, alpha_l1=None, alpha_l1_epoch_anneal=None,
                 alpha_l1_anneal_each=5, gamma_ext=None, gamma_epoch_anneal=None, gamma_anneal_each=5, beta=1.,
                 print_stats=False, **kwargs):
        super().__init__(model, adata, **kwargs)
        self.print_stats = print_stats
        self.alpha = alpha
        self.omega = omega
        if self.omega is not None:
            self.omega = self.omega.to(self.device)
        self.gamma_ext = gamma_ext
        self.gamma_epoch_anneal = gamma_epoch_anneal
        self.gamma_anneal_each = gamma_anneal_each
        self.alpha_l1 = alpha_l1
        self.alpha_l1_epoch_anneal = alpha_l1_epoch_anneal
        self.alpha_l1_anneal_each = alpha_l1_anneal_each
        self.beta = beta if self.model.use_hsic else None
        self.model.hsic_beta = float(beta) if self.beta is not None else 0.0
        self.watch_lr = None
        self.use_prox_ops = self.check_prox_ops()
        self.prox_ops = {}
        self.corr_coeffs = self.init_anneal()

    # ------------------------------------------------------------------------------------------ configuration
    def check_prox_ops(self):
        L0 = self.model.decoder.L0
        use_main = L0.expr_L.weight.requires_grad
        use = {'main_group_lasso': use_main and self.alpha is not None,
               'main_soft_mask': use_main and self.model.mask is not None and self.alpha_l1 is not None}
        use_ext = self.model.n_ext_decoder > 0 and self.gamma_ext is not None
        use['ext_unannot_l1'] = bool(use_ext and L0.ext_L.weight.requires_grad)
        use_ext_m = self.model.n_ext_m_decoder > 0 and self.alpha_l1 is not None
        use_ext_m = use_ext_m and L0.ext_L_m.weight.requires_grad
        use['ext_soft_mask'] = bool(use_ext_m and self.model.ext_mask is not None)
        return use

    def init_anneal(self):
        coeffs = {}
        soft = self.use_prox_ops['main_soft_mask'] or self.use_prox_ops['ext_soft_mask']
        coeffs['alpha_l1'] = 1. / self.alpha_l1_epoch_anneal if soft and self.alpha_l1_epoch_anneal is not None else 1.
        coeffs['gamma_ext'] = (1. / self.gamma_epoch_anneal
                               if self.use_prox_ops['ext_unannot_l1'] and self.gamma_epoch_anneal is not None else 1.)
        return coeffs

    def anneal(self):
        any_change = False
        if self.corr_coeffs['gamma_ext'] < 1.:
            any_change = True
            if self.epoch > 0 and self.epoch % self.gamma_anneal_each == 0:
                self.corr_coeffs['gamma_ext'] = min(self.epoch / self.gamma_epoch_anneal, 1.)
                if self.print_stats:
                    print('New gamma_ext anneal coefficient:', self.corr_coeffs['gamma_ext'])
        if self.corr_coeffs['alpha_l1'] < 1.:
            any_change = True
            # (the reference spells this attribute `self.self.alpha_l1_anneal_each`, an AttributeError at run time;
            #  the intended schedule is implemented here)
            if self.epoch > 0 and self.epoch % self.alpha_l1_anneal_each == 0:
                self.corr_coeffs['alpha_l1'] = min(self.epoch / self.alpha_l1_epoch_anneal, 1.)
                if self.print_stats:
                    print('New alpha_l1 anneal coefficient:', self.corr_coeffs['alpha_l1'])
        return any_change

    # ------------------------------------------------------------------------------------------ proximal operators
    def init_prox_ops(self):
        if any(self.use_prox_ops.values()) and self.watch_lr is None:
            self.watch_lr = self.optimizer.param_groups[0]['lr']
        if 'main_group_lasso' not in self.prox_ops and self.use_prox_ops['main_group_lasso']:
            print('Init the group lasso proximal operator for the main terms.')
            self.prox_ops['main_group_lasso'] = ProxGroupLasso(self.alpha * self.watch_lr, self.omega)
        if 'main_soft_mask' not in self.prox_ops and self.use_prox_ops['main_soft_mask']:
            print('Init the soft mask proximal operator for the main terms.')
            a = self.alpha_l1 * self.watch_lr * self.corr_coeffs['alpha_l1']
            self.prox_ops['main_soft_mask'] = ProxL1(a, self.model.mask.to(self.device))
        if 'ext_unannot_l1' not in self.prox_ops and self.use_prox_ops['ext_unannot_l1']:
            print('Init the L1 proximal operator for the unannotated extension.')
            self.prox_ops['ext_unannot_l1'] = ProxL1(self.gamma_ext * self.watch_lr * self.corr_coeffs['gamma_ext'])
        if 'ext_soft_mask' not in self.prox_ops and self.use_prox_ops['ext_soft_mask']:
            print('Init the soft mask proximal operator for the annotated extension.')
            a = self.alpha_l1 * self.watch_lr * self.corr_coeffs['alpha_l1']
            self.prox_ops['ext_soft_mask'] = ProxL1(a, self.model.ext_mask.to(self.device))

    def update_prox_ops(self):
        if 'main_group_lasso' in self.prox_ops:
            a = self.alpha * self.watch_lr
            if self.prox_ops['main_group_lasso']._alpha != a:
                self.prox_ops['main_group_lasso'] = ProxGroupLasso(a, self.omega)
        if 'ext_unannot_l1' in self.prox_ops:
            self.prox_ops['ext_unannot_l1']._alpha = self.gamma_ext * self.watch_lr * self.corr_coeffs['gamma_ext']
        for key in ('main_soft_mask', 'ext_soft_mask'):
            if key in self.prox_ops:
                self.prox_ops[key]._alpha = self.alpha_l1 * self.watch_lr * self.corr_coeffs['alpha_l1']

    def apply_prox_ops(self):
        L0 = self.model.decoder.L0
        if 'main_soft_mask' in self.prox_ops:
            self.prox_ops['main_soft_mask'](L0.expr_L.weight.data)
        if 'main_group_lasso' in self.prox_ops:
            self.prox_ops['main_group_lasso'](L0.expr_L.weight.data)
        if 'ext_unannot_l1' in self.prox_ops:
            self.prox_ops['ext_unannot_l1'](L0.ext_L.weight.data)
        if 'ext_soft_mask' in self.prox_ops:
            self.prox_ops['ext_soft_mask'](L0.ext_L_m.weight.data)

    # ------------------------------------------------------------------------------------------ loop hooks
    def on_iteration(self, batch_data):
        self.init_prox_ops()
        super().on_iteration(batch_data)
        self.apply_prox_ops()

    def check_early_stop(self):
        continue_training = super().check_early_stop()
        if continue_training:
            new_lr = self.optimizer.param_groups[0]['lr']
            if self.watch_lr is not None and self.watch_lr != new_lr:
                self.watch_lr = new_lr
                self.update_prox_ops()
        return continue_training

    def on_epoch_end(self):
        if self.print_stats:
            L0 = self.model.decoder.L0
            if self.use_prox_ops['main_group_lasso']:
                msg = f'Number of deactivated terms: {self.model.decoder.n_inactive_terms()}'
                print(('\n' + msg) if self.epoch > 0 else msg)
                print('-------------------')
            if self.use_prox_ops['main_soft_mask']:
                off = self.prox_ops['main_soft_mask']._I == 0
                share = ((L0.expr_L.weight.data.abs() == 0) & off).float().sum().item() / self.model.n_inact_genes
                print('Share of deactivated inactive genes: %.4f' % share)
                print('-------------------')
            if self.use_prox_ops['ext_soft_mask']:
                off = self.prox_ops['ext_soft_mask']._I == 0
                share = ((L0.ext_L_m.weight.data.abs() == 0) & off).float().sum().item() / self.model.n_inact_ext_genes
                print('Share of deactivated inactive genes in extension terms: %.4f' % share)
                print('-------------------')
            if self.use_prox_ops['ext_unannot_l1']:
                active = (L0.ext_L.weight.data.abs() > 0).sum(0).cpu().numpy()
                print('Active genes in unannotated extension terms:', active)
                print('Sparcity share in unannotated extension terms:', 1. - active / self.model.input_dim)
                print('-------------------')
        if self.anneal():
            self.update_prox_ops()
        super().on_epoch_end()

    def _read_logs(self, prefix):
        """loss = recon + alpha*kl + beta*hsic; unweighted = recon + kl + hsic (regularized.py:345-361).  The engine
        accumulates the weighted HSIC term in the third loss slot."""
        acc = self.engine.accum.clone()
        if self.world > 1:
            from ... import distributed as sdist
            sdist.allreduce_sum_(acc)
        acc = acc.cpu().numpy()
        n = max(acc[4], 1.0)
        recon, kl, whs, tot = acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n
        hs = whs / self.beta if (self.beta is not None and self.beta != 0) else 0.0
        self.logs[prefix + "loss"].append(tot)
        self.logs[prefix + "unweighted_loss"].append(recon + kl + hs)
        self.logs[prefix + "recon_loss"].append(recon)
        self.logs[prefix + "kl_loss"].append(kl)
        if self.beta is not None and self.model.use_hsic:
            self.logs[prefix + "hsic_loss"].append(hs)
