"""Hierarchical Gaussian-mixture clustering of the weighted history (reference: tempest/cluster.py).
Placeholder until the device E/M-step lands."""


class HierarchicalGaussianMixture:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        raise NotImplementedError(
            "clustering=True is not implemented on the GPU path yet: pass clustering=False "
            "(single global proposal mode, tempest/steps/train.py:118-122)")
