"""subsample_labels (reference lvc/modeling/sampling.py:9-57): index plumbing with torch.randperm (the RNG stream is
torch's, as in the reference, so `torch.manual_seed` / monkey-patched `randperm` reproduce a given sample)."""
import torch


def subsample_labels(labels, num_samples, positive_fraction, bg_label, inference=False):
    positive = ((labels != -1) & (labels != bg_label)).nonzero(as_tuple=True)[0]
    negative = (labels == bg_label).nonzero(as_tuple=True)[0]
    if inference:
        return positive, negative
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[perm1], negative[perm2]
