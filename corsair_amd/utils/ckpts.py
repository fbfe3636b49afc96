"""Checkpoint format of the reference (utils/ckpts.py:21-63): one ``torch.save`` dict with the entries
``state_dict`` (network), ``embedding_state_dict`` (descriptor head, optional), ``optimizer``,
``scheduler``, ``epoch``.  Same function names, argument order and return values; the state-dict
names are the reference's, so its released checkpoints load unchanged."""
import os

import torch


def load_checkpoint(model, embedding, optimizer, scheduler, path, trust_pickle=False):
    """utils/ckpts.py:21-40.  Optimizer / scheduler state of a training checkpoint may hold arbitrary
    pickled objects: `trust_pickle=True` opts in to full unpickling for files from a trusted source;
    the default loads tensors and plain containers only (released checkpoints are third-party files)."""
    checkpoint = torch.load(path, map_location="cpu", weights_only=not trust_pickle)
    model.load_state_dict(checkpoint["state_dict"])
    embedding.load_state_dict(checkpoint["embedding_state_dict"])
    optimizer.load_state_dict(checkpoint["optimizer"])
    scheduler.load_state_dict(checkpoint["scheduler"])
    epoch = checkpoint["epoch"]
    return model, embedding, optimizer, epoch


def save_checkpoint(model, embedding, optimizer, scheduler, epoch, save_dir, save_name):
    if not os.path.exists(save_dir):
        os.mkdir(save_dir)
    path = os.path.join(save_dir, save_name)
    state = {"state_dict": model.state_dict()}
    if embedding is not None:
        state["embedding_state_dict"] = embedding.state_dict()
    state.update(optimizer=optimizer.state_dict(), scheduler=scheduler.state_dict(), epoch=epoch)
    torch.save(state, path)


def load_state_dicts(path):
    """(state_dict, embedding_state_dict or None) of a checkpoint file as host tensors: what
    ``corsair_amd.engine.ResUNetEngine`` consumes (evaluation.py:195-201 loads the same two entries)."""
    checkpoint = torch.load(path, map_location="cpu", weights_only=True)   # two dicts of tensors: no pickle code
    return checkpoint["state_dict"], checkpoint.get("embedding_state_dict")
