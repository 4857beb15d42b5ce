"""Checkpoint interchange with the reference's dict format (utils/io_utils.py:37-105).

`save_checkpoint` writes the reference's keys -- epoch, model_type, optimizer, model_state, optimizer_state, cg -- under
the reference's file name `<ckptdir>/<dataset>_<method>_h<hidden>_o<out>/[best|<epoch>].pth.tar`; `load_checkpoint` reads
files either side wrote.  The reference pickles the optimizer OBJECT under "optimizer"; here that entry is kept for a
torch optimizer and is None for a `TrainEngine` (whose moments, step count and hyper-parameters are a plain tensor dict
under "optimizer_state": `TrainEngine.state_dict()`), so training can be resumed in both loops."""
from __future__ import annotations

import os
from typing import Optional

import torch


def gen_prefix(args) -> str:
    """io_utils.py:37-53"""
    name = args.bmname if getattr(args, "bmname", None) is not None else args.dataset
    return f"{name}_{args.method}_h{args.hidden_dim}_o{args.output_dim}"


def create_filename(save_dir, args, isbest: bool = False, num_epochs: int = -1) -> str:
    """io_utils.py:65-80 (note `num_epochs > 0`: epoch 0 and -1 both map to '<prefix>.pth.tar')"""
    filename = os.path.join(save_dir, gen_prefix(args))
    os.makedirs(filename, exist_ok=True)
    if isbest:
        filename = os.path.join(filename, "best")
    elif num_epochs > 0:
        filename = os.path.join(filename, str(num_epochs))
    return filename + ".pth.tar"


def save_checkpoint(model, optimizer, args, num_epochs: int = -1, isbest: bool = False, cg_dict=None) -> str:
    """io_utils.py:83-105.  `optimizer` is a torch optimizer (reference-style loop) or a `TrainEngine`."""
    from ..model import TrainEngine
    filename = create_filename(args.ckptdir, args, isbest, num_epochs=num_epochs)
    is_engine = isinstance(optimizer, TrainEngine)
    torch.save({"epoch": num_epochs, "model_type": args.method, "optimizer": None if is_engine else optimizer,
                "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(), "cg": cg_dict}, filename)
    return filename


def load_checkpoint(path: str, model=None, optimizer=None, map_location: Optional[str] = None, trusted: bool = False) -> dict:
    """Read a dict checkpoint (the reference's `load_ckpt`, io_utils.py:108-129, returns the dict and leaves the rest to
    the caller) and, when given, restore `model` (state_dict keys are the reference's) and `optimizer` (a torch
    optimizer or a `TrainEngine`; a reference-written Adam state is converted for the engine by parameter order).
    A bare state_dict file (train_chargrid_funsd_msau.py:100-102 `torch.save(model.state_dict(), ...)`) is accepted too.
    `trusted=True` allows the full unpickler for reference-written files (see below)."""
    from ..model import TrainEngine
    if not os.path.isfile(path):
        raise FileNotFoundError(f"checkpoint {path!r} does not exist")
    if map_location is None and model is not None:
        map_location = str(model.flat_parameters.device)
    # Tensor-only files -- bare state_dicts and what save_checkpoint writes for a TrainEngine (optimizer = None) -- load with
    # weights_only=True: no pickle code runs.  The reference's save_checkpoint also pickles the torch optimizer OBJECT
    # (utils/io_utils.py:93-97); reading that needs the full unpickler, i.e. it executes whatever the file contains, and is
    # therefore opt-in: trusted=True, for files you wrote yourself.
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:                       # pickle.UnpicklingError and friends: not a tensor-only file
        if not trusted:
            raise RuntimeError(f"{path!r} is not a tensor-only checkpoint (it pickles Python objects, as the reference's "
                               f"save_checkpoint does for its optimizer); pass trusted=True to load_checkpoint to unpickle "
                               f"it -- only for files from a source you trust") from e
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if "model_state" not in ckpt:
        ckpt = {"epoch": -1, "model_type": None, "optimizer": None, "model_state": ckpt, "optimizer_state": None, "cg": None}
    if model is not None:
        model.load_state_dict(ckpt["model_state"])
    if optimizer is not None and ckpt.get("optimizer_state") is not None:
        st = ckpt["optimizer_state"]
        if isinstance(optimizer, TrainEngine):
            optimizer.load_state_dict(st)
        else:
            optimizer.load_state_dict(st)
    return ckpt
