"""Reading and writing the reference's checkpoint files (nerf/utils.py: Trainer.save_checkpoint :1083-1143, load_checkpoint :1145-1205).

The models here carry the reference's parameter and buffer names, shapes and dtypes (pinned by tests/golden/state_dict_layout.json,
generated from the reference's own modules), so a `.pth` written by the reference trainer loads unchanged and vice versa.
Only the model part is handled: optimizer / scheduler / scaler / EMA state belong to the trainer, which is out of scope.
"""
import os

import torch


def read(checkpoint, map_location="cpu"):
    """Path or already-loaded object -> (model_state_dict, meta dict).  A file holding a bare state_dict (the reference accepts
    those: utils.py:1157-1160) comes back with meta['bare'] = True."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=False)
    if "model" not in checkpoint:
        return checkpoint, {"bare": True}
    meta = {k: checkpoint[k] for k in ("epoch", "global_step", "stats", "mean_count", "mean_density") if k in checkpoint}
    meta["bare"] = False
    return checkpoint["model"], meta


def load_model(model, checkpoint, map_location="cpu", log=None):
    """Same behaviour as the model part of Trainer.load_checkpoint: a bare state_dict loads strictly, a trainer checkpoint with
    strict=False (the 'best' files drop density_grid, utils.py:1135; a PaletteNetwork is initialised from a NeRF checkpoint the
    same way) and restores mean_count / mean_density of cuda_ray models.  Returns {'missing', 'unexpected', **meta}."""
    state, meta = read(checkpoint, map_location)
    if meta["bare"]:
        model.load_state_dict(state)
        missing, unexpected = [], []
    else:
        res = model.load_state_dict(state, strict=False)
        missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
        if getattr(model, "cuda_ray", False):
            if "mean_count" in meta:
                model.mean_count = meta["mean_count"]
            if "mean_density" in meta:
                model.mean_density = meta["mean_density"]
    if log:
        log("[INFO] loaded model.")
        if missing:
            log(f"[WARN] missing keys: {missing}")
        if unexpected:
            log(f"[WARN] unexpected keys: {unexpected}")
    # load_state_dict copies in place, which bumps the tensors' versions: the packed MFMA weights and the occupancy mip are
    # keyed on those versions and rebuild themselves on the next frame.
    return dict(meta, missing=missing, unexpected=unexpected)


def save_model(model, path, epoch=0, global_step=0, stats=None, best=False):
    """Write a file the reference's load_checkpoint reads.  best=True mirrors its '<name>.pth' files: no density_grid."""
    state = {"epoch": epoch, "global_step": global_step, "stats": stats if stats is not None else {}}
    if getattr(model, "cuda_ray", False):
        state["mean_count"] = model.mean_count
        state["mean_density"] = model.mean_density
    sd = model.state_dict()
    if best and "density_grid" in sd:
        sd = {k: v for k, v in sd.items() if k != "density_grid"}
    state["model"] = sd
    torch.save(state, path)
    return path
