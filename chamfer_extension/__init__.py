"""Reference import path `chamfer_extension.dist_chamfer`."""
