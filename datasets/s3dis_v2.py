from cloud_transformers_amd.data.datasets import (ChromaticAutoContrast, ChromaticJitter, ChromaticTranslation,  # noqa: F401
                                                  HueSaturationTranslation, Indoor3DSemSeg, RandomJitter, RandomRotate,
                                                  RandomScale, RandomSymmetries)
