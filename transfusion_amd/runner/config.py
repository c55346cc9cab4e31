"""Run-config loader with the reference's YAML surface (runner/run_experiment.py:68-151 ``update_config`` and
runner/utils/envyaml_wrapper.py:4-18): ``${CODE}/${DATA}/${RUNS}`` expansion, the composition of the run YAML with
the fusion YAML (``run.narr_fusion.config`` merged into ``run.narr_fusion``) and the derived keys the fusion block
depends on (``text_pooling``, ``input_f_size`` <- ``out_mlp``, ``size``).  Everything outside the fusion path
(wandb, datasets, detector YAML) is left untouched in the returned dict.
"""
from __future__ import annotations

import os
import re

import yaml

LANG_MODEL_FEATURE_SIZES = {"all-MiniLM-L12-v2": 384}          # run_experiment.py:43-51 (the entry the shipped YAMLs use)
LM_TO_TEXT_POOLING = {"all-MiniLM-L12-v2": "sbert_finetune"}   # run_experiment.py:53-60
LEARNABLE_LM = {"sbert_finetune", "gpt2", "t5-wikihow", "slowfast"}


def _expand(node):
    if isinstance(node, dict):
        return {k: _expand(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_expand(v) for v in node]
    if isinstance(node, str):
        return re.sub(r"\$\{(\w+)\}", lambda m: os.environ.get(m.group(1), m.group(0)), node)
    return node


def load_yaml(path: str) -> dict:
    with open(os.path.expandvars(path)) as f:
        return _expand(yaml.safe_load(f))


def load_fusion_config(path: str) -> dict:
    return load_yaml(path)


def update_config(config: dict) -> dict:
    """The fusion-relevant part of run_experiment.py:update_config (lines cited inline)."""
    run = config["run"]
    run["narr_fusion"].update(load_fusion_config(run["narr_fusion"]["config"]))                    # :75-77
    args = run["narration_embeds"]["args"]
    args["text_pooling"] = LM_TO_TEXT_POOLING.get(args["model_v"], args["model_v"])                 # :87-89
    if run["narration_embeds"].get("slowfast_f", False):                                            # :90-92
        args["text_pooling"] = "slowfast"
        args["model_v"] = "slowfast"
    if args.get("pooling") == "sbert" or args["text_pooling"] in LEARNABLE_LM:                      # :94-121
        if args["out_mlp"]:
            run["narr_fusion"]["args"]["input_f_size"] = args["out_mlp"]                            # :99-100
            args["size"] = LANG_MODEL_FEATURE_SIZES.get(args["model_v"], args["size"])
        else:
            run["narr_fusion"]["args"]["input_f_size"] = LANG_MODEL_FEATURE_SIZES.get(args["model_v"], args["size"])
            args["size"] = run["narr_fusion"]["args"]["input_f_size"]
    else:
        run["narr_fusion"]["args"]["input_f_size"] = args["size"]                                   # :122-123
    if args["text_pooling"] in LEARNABLE_LM:
        args["finetune"] = False                                                                     # :126-127
    run["experiment"] = config["experiment"]                                                        # :150
    return config
