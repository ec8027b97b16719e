"""Checkpoint-on-improvement (volume_segmantics/utilities/early_stopping.py:7-63): the ``.pytorch`` wire format
{model_state_dict, model_struc_dict, optimizer_state_dict, loss_val, label_codes} is kept so files interchange."""
import logging

import numpy as np
import torch


class EarlyStopping:
    ENGINE_KEYS = ("precision", "allow_random_encoder")   # not smp.Unet keyword arguments: the reference does smp.Unet(**struct)

    def __init__(self, patience=7, verbose=False, delta=0, path="checkpoint.pt", model_dict=None, best_score=None):
        self.patience, self.verbose, self.delta, self.path = patience, verbose, delta, path
        model_dict = dict(model_dict) if model_dict is not None else {}
        self.engine_settings = {k: model_dict.pop(k) for k in self.ENGINE_KEYS if k in model_dict}
        self.model_struc_dict = model_dict
        self.best_score = best_score
        self.val_loss_min = np.inf if best_score is None else -best_score
        self.counter, self.early_stop = 0, False
        self.write = True    # data-parallel runs: every rank takes the decisions (on the same reduced loss), rank 0 writes the file

    def __call__(self, val_loss, model, optimizer, label_codes):
        score = -val_loss
        if self.best_score is not None and score < self.best_score + self.delta:
            self.counter += 1
            logging.info(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            self.early_stop = self.counter >= self.patience
            return
        self.best_score, self.counter = score, 0
        self.save_checkpoint(val_loss, model, optimizer, label_codes)

    def save_checkpoint(self, val_loss, model, optimizer, label_codes):
        if not self.write:
            self.val_loss_min = val_loss
            return
        if self.verbose:
            logging.info(f"Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...")
        ckpt = {"model_state_dict": model.state_dict(), "model_struc_dict": self.model_struc_dict,
                "optimizer_state_dict": optimizer.state_dict(), "loss_val": val_loss, "label_codes": label_codes}
        if self.engine_settings:   # an extra key the reference never reads; absent = exactly the reference's five keys
            ckpt["engine_settings"] = self.engine_settings
        torch.save(ckpt, self.path)
        self.val_loss_min = val_loss
