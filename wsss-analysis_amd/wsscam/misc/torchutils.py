"""misc.torchutils helpers of the CAM path (upstream jiwoon-ahn/irn semantics, SURVEY.md
Appendix A).  Call site: 03b_irn/step/make_cam.py:120."""
import numpy as np


class Subset:
    """torch.utils.data.Subset without the torch dependency."""

    def __init__(self, dataset, indices):
        self.dataset = dataset
        self.indices = indices

    def __getitem__(self, idx):
        return self.dataset[int(self.indices[idx])]

    def __len__(self):
        return len(self.indices)


def split_dataset(dataset, n_splits):
    """Round-robin shards: GPU g of G owns images[g::G]."""
    return [Subset(dataset, np.arange(i, len(dataset), n_splits)) for i in range(n_splits)]


def gap2d(x, keepdims=False):
    out = x.reshape(x.shape[0], x.shape[1], -1).mean(-1)
    if keepdims:
        out = out.reshape(out.shape[0], out.shape[1], 1, 1)
    return out
