"""Reference import path `unet2d.unet_parts` (grouped 2D conv blocks)."""
