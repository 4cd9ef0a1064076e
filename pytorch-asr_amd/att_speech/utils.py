"""att_speech.utils — the registry functions of the reference
(att_speech/utils.py:73-97): YAML `class_name` -> object."""
from __future__ import absolute_import, division, print_function

import importlib


def get_class(str_or_class, default_mod=None):
    if isinstance(str_or_class, str):
        parts = str_or_class.split('.')
        mod_name = '.'.join(parts[:-1])
        class_name = parts[-1]
        if mod_name:
            mod = importlib.import_module(mod_name)
        elif default_mod is not None:
            mod = importlib.import_module(default_mod)
        else:
            raise ValueError('Specify a module for %s' % (str_or_class,))
        return getattr(mod, class_name)
    return str_or_class


def contruct_from_kwargs(object_kwargs, default_mod=None,
                         additional_parameters=None):
    object_kwargs = dict(object_kwargs)
    class_name = object_kwargs.pop('class_name')
    klass = get_class(class_name, default_mod)
    if additional_parameters:
        object_kwargs.update(additional_parameters)
    return klass(**object_kwargs)


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
