"""Shared model utilities (reference: promonet/model/core.py)."""
import torch


def get_padding(kernel_size, dilation=1, stride=1):
    """'Same' padding of a dilated conv (promonet/model/core.py:9-11)."""
    return int((kernel_size * dilation - dilation - stride + 1) / 2)


def attach(root, dotted_key, tensor, buffer=False):
    """Register `tensor` on `root` under a dotted state-dict key, creating
    bare container modules on the way, so `root.state_dict()` carries exactly
    the reference's key names (the checkpoint contract)."""
    *path, leaf = dotted_key.split('.')
    module = root
    for name in path:
        if name not in module._modules:
            module.add_module(name, torch.nn.Module())
        module = module._modules[name]
    if buffer:
        module.register_buffer(leaf, tensor)
    else:
        module.register_parameter(
            leaf, torch.nn.Parameter(tensor, requires_grad=False))
