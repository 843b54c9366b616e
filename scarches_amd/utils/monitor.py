"""Early stopping / LR-on-plateau state machine with the reference's semantics
(scarches/utils/monitor.py:32-115): it is fed one scalar per epoch; the first `patience - 1` calls always
continue; afterwards it tracks the best value, counts epochs without an improvement larger than `threshold`,
asks for a learning-rate cut every `lr_patience` such epochs and stops after `patience` of them.
`update_state` keeps a separate best for checkpointing decisions.
"""
import numpy as np


class EarlyStopping(object):
    def __init__(self,
                 early_stopping_metric: str = "val_unweighted_loss",
                 mode: str = "min",
                 threshold: float = 0,
                 patience: int = 20,
                 reduce_lr: bool = True,
                 lr_patience: int = 13,
                 lr_factor: float = 0.1):
        self.early_stopping_metric = early_stopping_metric
        self.mode = mode
        self.threshold = threshold
        self.patience = patience
        self.reduce_lr = reduce_lr
        self.lr_patience = lr_patience
        self.lr_factor = lr_factor

        self.epoch = 0
        self.wait = 0
        self.wait_lr = 0
        self.current_performance = np.inf
        worst = np.inf if mode == "min" else -np.inf
        self.best_performance = worst
        self.best_performance_state = worst
        self._disabled = patience == 0      # reference: patience 0 turns step() into "stop immediately"

    def _gain(self, best, value):
        return best - value if self.mode == "min" else value - best

    def step(self, scalar):
        """Returns (continue_training, update_lr)."""
        if self._disabled:
            return False
        self.epoch += 1
        if self.epoch < self.patience:
            return True, False
        if self.wait >= self.patience:
            print("\nStopping early: no improvement of more than " + str(self.threshold) +
                  " nats in " + str(self.patience) + " epochs")
            print("If the early stopping criterion is too strong, "
                  "please instantiate it with different parameters in the train method.")
            return False, False
        update_lr = False
        if self.reduce_lr and self.wait_lr >= self.lr_patience:
            update_lr = True
            self.wait_lr = 0
        self.current_performance = scalar
        gain = self._gain(self.best_performance, scalar)
        if gain > 0:
            self.best_performance = scalar
        if gain < self.threshold:
            self.wait += 1
            self.wait_lr += 1
        else:
            self.wait = 0
            self.wait_lr = 0
        return True, update_lr

    def update_state(self, scalar):
        improved = self._gain(self.best_performance_state, scalar) > 0
        if improved:
            self.best_performance_state = scalar
        return improved
