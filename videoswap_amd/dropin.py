"""Run a script written against the reference's import surface — in particular the reference's own `test.py`,
UNCHANGED — on this package:

    cd <workspace with datasets/ experiments/ options/>            # the YAMLs use paths relative to the cwd
    python -m videoswap_amd.dropin /root/reference/test.py -opt options/test_videoswap/.../x.yml

The shim packages under videoswap_amd/shims (`diffusers`, `omegaconf`, `videoswap.{data,models,pipelines,utils}`)
are put in front of `sys.path`, so `from videoswap.models import build_model`, `from diffusers import DDIMScheduler`,
`from omegaconf import OmegaConf` ... resolve to the HIP-backed classes; `accelerate` is the installed package."""
import os
import runpy
import sys

SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')


def install():
    if SHIMS not in sys.path:
        sys.path.insert(0, SHIMS)
    for name in list(sys.modules):          # anything already imported under the shimmed names must come from the shims
        if name.split('.')[0] in ('videoswap', 'diffusers', 'omegaconf'):
            if not (getattr(sys.modules[name], '__file__', None) or '').startswith(SHIMS):
                del sys.modules[name]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit('usage: python -m videoswap_amd.dropin <script.py> [script args]')
    script = argv[0]
    install()
    sys.argv = [script] + argv[1:]
    sys.dont_write_bytecode = True            # never write __pycache__ next to a read-only script
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
