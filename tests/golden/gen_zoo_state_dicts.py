#!/usr/bin/env python3
"""Record the state-dict layout (parameter / buffer names and shapes — data, not source) of every model in the
reference's model_zoo, built on the REFERENCE's own layers, into tests/golden/zoo_state_dicts.json.

Runs only in the build container (needs /root/reference); uses the same stand-ins for the two third-party modules as
gen_golden.py.  tests/test_zoo_cpu.py then builds the same model files on THIS package's layers and compares.

usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_zoo_state_dicts.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402  (stand-ins; puts /root/reference on sys.path)

ZOO = ["model_zoo/s3dis/segmenter.py", "model_zoo/s3dis/segmenter_pad.py", "model_zoo/scanobject/classifier.py",
       "model_zoo/scanobject/classifier_scales.py", "model_zoo/completion/inpainter.py",
       "model_zoo/image_reconstruction/reconstructor.py"]


def main():
    G._install_shims()
    if G.REF not in sys.path:
        sys.path.insert(0, G.REF)
    out = {}
    for rel in ZOO:
        ns = {"__name__": "zoo_model"}
        try:
            with open(os.path.join(G.REF, rel)) as f:
                exec(compile(f.read(), rel, "exec"), ns)      # what utils/train_util.py:23-34 does with a model file
            model = ns["Model"]()
        except Exception as ex:                              # noqa: BLE001 — the reconstructor needs torchvision (not installed)
            out[rel] = {"__error__": "%s: %s" % (type(ex).__name__, ex)}
            print(rel, "skipped:", out[rel]["__error__"])
            continue
        out[rel] = {k: list(v.shape) for k, v in model.state_dict().items()}
        print(rel, len(out[rel]), "tensors")
    with open(os.path.join(HERE, "zoo_state_dicts.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
