"""Fit-loop utilities with the reference's names and on-disk formats (reference utils.py:23-88)."""
from __future__ import annotations

import random

import numpy as np
import torch


def seed(seed: int = 0):
    """reference utils.py:77-88"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def save_checkpoint(model, optimizer, filename, learning_rate=None, epoch=None):
    """reference utils.py:23-32: same dict keys, torch-format optimizer state."""
    print('Saving Checkpoint...')
    checkpoint = {
        'model_state_dict': model.state_dict(),
        'optimizer_state_dict': optimizer.state_dict(),
        'learning_rate': learning_rate,
        'epoch': epoch,
    }
    torch.save(checkpoint, filename)


def load_checkpoint(checkpoint_file, model, optimizer=None, lr=None, map_location=None):
    """reference utils.py:34-50: restores weights (+ Adam moments) and overrides the LR with the CLI
    value; epoch counter and scheduler are NOT restored (reference behaviour, Appendix A-8)."""
    checkpoint = torch.load(checkpoint_file, map_location=map_location, weights_only=True)
    print('Number of Epochs: ', checkpoint['epoch'])
    print('Learning Rate: ', checkpoint['learning_rate'])
    model.load_state_dict(checkpoint['model_state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        if lr is not None:
            for param_group in optimizer.param_groups:
                param_group['lr'] = lr
        elif checkpoint['learning_rate'] is not None:
            new_lr = checkpoint['learning_rate']
            if isinstance(new_lr, (list, tuple)):      # the reference stores scheduler.get_last_lr() (a list)
                new_lr = new_lr[0]
            for param_group in optimizer.param_groups:
                param_group['lr'] = new_lr
    return checkpoint


def shard_indices(n: int, epoch: int, rank: int, world: int, batch_size: int, shuffle: bool = True, seed: int = 0):
    """Batch-sharded sampler: an epoch-seeded permutation (the reference's DataLoader(shuffle=True),
    train.py:67) cut into global batches of world*batch_size; rank r takes slice r of each.  With one
    rank the ragged tail batch is kept, as the reference's DataLoader (drop_last=False) keeps it; with
    several ranks it is dropped so that every rank runs the same number of steps (the gradient
    all-reduce is a collective).  Returns a list of index arrays, one per step."""
    rng = np.random.default_rng(seed + epoch)
    order = rng.permutation(n) if shuffle else np.arange(n)
    gb = world * batch_size
    steps = n // gb
    out = [order[s * gb + rank * batch_size: s * gb + (rank + 1) * batch_size] for s in range(steps)]
    if world == 1 and n % gb:
        out.append(order[steps * gb:])
    return out
