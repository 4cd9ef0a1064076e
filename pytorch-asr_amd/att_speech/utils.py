"""att_speech.utils — the registry functions of the reference
(att_speech/utils.py:73-97): YAML `class_name` -> object."""
from __future__ import absolute_import, division, print_function

import importlib


def get_class(str_or_class, default_mod=None):
    """`'pkg.module.Name'` -> the object; a bare `'Name'` is looked up in `default_mod`;
    anything that is not a string is returned as it is."""
    if not isinstance(str_or_class, str):
        return str_or_class
    module_path, _, attribute = str_or_class.rpartition('.')
    module_path = module_path or default_mod
    if not module_path:
        raise ValueError('Specify a module for %s' % (str_or_class,))
    return getattr(importlib.import_module(module_path), attribute)


def contruct_from_kwargs(object_kwargs, default_mod=None,
                         additional_parameters=None):
    """Instantiate `object_kwargs['class_name']` with the remaining entries as keyword
    arguments; `additional_parameters` are added last (they win).  The name keeps the
    reference's spelling because callers import it."""
    kwargs = {k: v for k, v in dict(object_kwargs).items() if k != 'class_name'}
    kwargs.update(additional_parameters or {})
    return get_class(object_kwargs['class_name'], default_mod)(**kwargs)


def edit_distance(x, y):
    """Levenshtein distance between two sequences (reference utils.py:18-33)."""
    prev = list(range(len(y) + 1))
    for i, xi in enumerate(x, 1):
        cur = [i]
        for j, yj in enumerate(y, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (xi != yj)))
        prev = cur
    return prev[-1]


def get_mask(lengths, mask_length=None, batch_first=True):
    """1 inside each sequence, 0 on padding (reference utils.py:436-448);
    always on the CPU like `lengths`."""
    import torch
    lengths = torch.as_tensor(lengths)
    if mask_length is None:
        mask_length = int(lengths.max())
    lengths = lengths.long()
    if batch_first:
        mask = torch.arange(mask_length) < lengths[:, None]
    else:
        mask = torch.arange(mask_length)[:, None] < lengths
    return mask.float()
