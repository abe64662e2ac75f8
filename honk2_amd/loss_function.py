"""Registered loss wrappers (reference ``loss_function.py:6-29``).  Only ``ce_loss`` is used by the shipped
configs; these are thin ``torch.nn.functional`` calls kept for config compatibility -- the fused on-GPU
evaluation tail (``kws_eval_batch``) computes the same cross-entropy without a per-batch sync."""
import torch.nn.functional as F

from .utils import register_cls


@register_cls('loss_fn.ce_loss')
def ce_loss(output, target):
    return F.cross_entropy(output, target.to(output.device))


@register_cls('loss_fn.nll_loss')
def nll_loss(output, target):
    return F.nll_loss(output, target.to(output.device))


@register_cls('loss_fn.bce_loss')
def bce_loss(output, target):
    return F.binary_cross_entropy(output, target.to(output.device))


@register_cls('loss_fn.logsoftmax_nll_loss')
def logsoftmax_nll_loss(output, target):
    return F.nll_loss(F.log_softmax(output, dim=1), target.to(output.device).max(1)[1].long())


@register_cls('loss_fn.softmax_bce_loss')
def softmax_bce_loss(output, target):
    return F.binary_cross_entropy(F.softmax(output, dim=1), target.to(output.device))


@register_cls('loss_fn.sigmoid_bce_loss')
def sigmoid_bce_loss(output, target):
    return F.binary_cross_entropy(output.sigmoid(), target.to(output.device))
