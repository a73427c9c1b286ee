#!/usr/bin/env python3
"""Write a one-line `vame` shim package into a directory on sys.path, so that scripts written for the reference
(`import vame; vame.train_model(config); vame.pose_segmentation(config)`, examples/demo.py:48,56) run on vame_amd unmodified.

    python tools/install_vame_alias.py <site-packages or any sys.path dir> [--force]

The shim only calls vame_amd.compat.install_alias(); remove the written `vame/` directory to undo.  It refuses to overwrite an
existing `vame` package (the reference itself) unless --force is given.
"""
import os
import sys

SHIM = ('"""vame -> vame_amd alias (written by vame_amd tools/install_vame_alias.py); see vame_amd/compat.py."""\n'
        "from vame_amd.compat import install_alias as _install\n"
        "_install(replace=True)\n")


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    target = os.path.join(os.path.abspath(sys.argv[1]), "vame")
    if os.path.exists(target) and "--force" not in sys.argv:
        raise SystemExit(f"{target} exists (the reference package?); pass --force to replace it")
    os.makedirs(target, exist_ok=True)
    with open(os.path.join(target, "__init__.py"), "w") as f:
        f.write(SHIM)
    print("wrote", os.path.join(target, "__init__.py"))


if __name__ == "__main__":
    main()
