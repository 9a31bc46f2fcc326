"""Frozen torchvision-style ResNet (BasicBlock) on the HIP conv kernels — the classifier of the StylEx step.

The classifier is frozen and in eval mode (reference stylex/resnet_classifier.py:29-71): its BatchNorms are
affine maps, so every `conv -> bn -> relu` / `conv -> bn -> (+identity) -> relu` of a BasicBlock is ONE fused
kernel launch (BN folded into the weights and a bias, residual merge and ReLU in the epilogue) with a data
gradient only (no parameter gradients exist).  Used in the 'bf16' speed mode; the fp32 parity mode keeps the
stock PyTorch module.  The 7x7 stem, max-pool, average pool and the linear head stay on PyTorch (3 % of the
FLOPs).
"""
import torch
from torch import nn

import ops


def _fold(conv, bn):
    """(W', b') with bn(conv(x, W)) == conv(x, W') + b' for an eval-mode BatchNorm."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = (conv.weight * scale.view(-1, 1, 1, 1)).detach().contiguous()
    b = (bn.bias - bn.running_mean * scale).detach().contiguous()
    if conv.bias is not None:
        b = b + conv.bias.detach() * scale
    return nn.Parameter(w, requires_grad=False), nn.Parameter(b, requires_grad=False)


class HipFrozenResNet(nn.Module):
    def __init__(self, model):
        super().__init__()
        assert not model.training, "the classifier must be in eval mode (BatchNorm folding)"
        self.model = model  # stem / pools / fc are used as they are
        self.folded = nn.ParameterList()
        self.plan = []  # per BasicBlock: indices into self.folded + strides
        for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
            for blk in layer:
                entry = {"stride": blk.conv1.stride[0]}
                for name, conv, bn in (("c1", blk.conv1, blk.bn1), ("c2", blk.conv2, blk.bn2)):
                    w, b = _fold(conv, bn)
                    entry[name] = len(self.folded)
                    self.folded.extend([w, b])
                if blk.downsample is not None:
                    w, b = _fold(blk.downsample[0], blk.downsample[1])
                    entry["down"] = len(self.folded)
                    self.folded.extend([w, b])
                self.plan.append(entry)

    @staticmethod
    def supports(model):
        try:
            blocks = [b for layer in (model.layer1, model.layer2, model.layer3, model.layer4) for b in layer]
            return all(hasattr(b, "conv1") and hasattr(b, "bn2") and not hasattr(b, "conv3") and
                       b.conv1.groups == 1 and b.conv1.kernel_size == (3, 3) for b in blocks)
        except AttributeError:
            return False

    def forward(self, x):
        m = self.model
        x = m.maxpool(m.relu(m.bn1(m.conv1(x))))
        x = x.to(ops.act_dtype()).contiguous(memory_format=torch.channels_last)
        prev = ops.set_fast(True)  # first-order gradients only ever flow through the frozen classifier
        try:
            for e in self.plan:
                w1, b1, w2, b2 = (self.folded[e["c1"]], self.folded[e["c1"] + 1], self.folded[e["c2"]],
                                  self.folded[e["c2"] + 1])
                idt = x
                if "down" in e:
                    idt = ops.conv2d(x, self.folded[e["down"]], self.folded[e["down"] + 1], stride=e["stride"], padding=0)
                out = ops.conv2d(x, w1, b1, stride=e["stride"], padding=1, lrelu="relu")
                x = ops.conv2d(out, w2, b2, stride=1, padding=1, lrelu="relu", residual=idt, res_scale=1.0)
        finally:
            ops.set_fast(prev)
        x = x.float()
        return m.fc(torch.flatten(m.avgpool(x), 1))
