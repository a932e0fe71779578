"""Run one of the reference's own drivers UNCHANGED on the MI355X-native hot path:

    cd <reference checkout> && python -m mvdfusion_amd.run demo.py -c configs/mvd_gso.yaml ...

installs the module aliases of mvdfusion_amd.configs.install_aliases() (the yaml `target:` strings and
`from utils.load_model import instantiate_from_config`, demo.py:21 / train.py:24, then resolve to this package's mirrors) and
executes the script as `__main__`.  No reference file is edited or copied.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__, file=sys.stderr)
        return 2
    script = argv[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or os.getcwd())
    from .configs import install_aliases
    install_aliases()
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
