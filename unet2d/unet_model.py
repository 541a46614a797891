from cloud_transformers_amd.layers.unet import UNet  # noqa: F401
