"""Focal cosine loss (https://arxiv.org/abs/2007.07805; reference losses/focal_cosine.py): cosine-embedding loss against
the one-hot target plus a focal cross entropy on L2-normalised logits.  ``[N, classes]`` classification input, plain
torch (runs on the MI355X through ATen)."""
import torch
import torch.nn.functional as F
from torch import Tensor, nn

__all__ = ["FocalCosineLoss"]


class FocalCosineLoss(nn.Module):
    def __init__(self, alpha: float = 1, gamma: float = 2, xent: float = 0.1, reduction="mean"):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma
        self.xent = xent
        self.reduction = reduction

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        onehot = F.one_hot(target, num_classes=input.size(-1))
        cosine = F.cosine_embedding_loss(input, onehot, torch.tensor([1], device=target.device), reduction=self.reduction)
        ce = F.cross_entropy(F.normalize(input), target, reduction="none")
        focal = self.alpha * (1 - torch.exp(-ce)) ** self.gamma * ce
        if self.reduction == "mean":
            focal = focal.mean()
        return cosine + self.xent * focal
