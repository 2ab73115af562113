"""Caffe2-style initialisers named as the reference uses them (fvcore.nn.weight_init, un-vendored;
call sites reference detectron2/modeling/backbone/resnet.py:190-193, fpn.py:73-74):
c2_xavier_fill = kaiming_uniform(a=1); c2_msra_fill = kaiming_normal(fan_out, relu); bias = 0."""
from torch import nn


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)
