"""Masked generative foresight head (reference mdt/models/img_generation/)."""
