from cloud_transformers_amd.layers.grouped_conv import Basic2DBlock, Res2DBlock  # noqa: F401
