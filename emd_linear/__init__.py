"""Reference import path `emd_linear.emd_module`."""
