from cloud_transformers_amd.emd import emdFunction, emdModule  # noqa: F401
