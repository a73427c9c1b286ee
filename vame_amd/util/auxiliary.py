"""config.yaml reader for the train / embed path.

Reference: vame/util/auxiliary.py:129-159 (`read_config`, ruamel.yaml).  Same contract -- returns the
mapping, raises FileNotFoundError with the reference's message for a missing file, and rewrites
`project_path` when the project folder moved (auxiliary.py:139-142) -- on PyYAML (ruamel is optional).
"""
import os
from pathlib import Path

import yaml

def read_config(configname):
    path = Path(configname)
    if not os.path.exists(path):
        raise FileNotFoundError(
            "Config file is not found. Please make sure that the file exists and/or that you passed the path of the "
            "config file correctly!")
    with open(path, "r") as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    curr_dir = os.path.dirname(str(configname))
    if cfg.get("project_path") != curr_dir:
        cfg["project_path"] = curr_dir
        write_config(configname, cfg)
    return cfg


def write_config(configname, cfg):
    with open(configname, "w") as cf:
        yaml.safe_dump(dict(cfg), cf, sort_keys=False, default_flow_style=False)
