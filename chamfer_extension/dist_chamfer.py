from cloud_transformers_amd.chamfer import (ChamferDist, ChamferFunction, loss_chamder_2d,  # noqa: F401
                                            loss_chamfer, loss_chamfer_adj)
