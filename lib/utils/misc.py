"""`NestedTensor`, the text container forward_test receives (reference lib/utils/misc.py:23-48): `.tensors` are the
token ids [B,T], `.mask` marks real tokens (1) vs padding (0)."""


class NestedTensor(object):
    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)
