"""Segmentation losses (focal / Dice / Jaccard / Lovasz) on HIP kernels -- populated in losses/*.py."""
