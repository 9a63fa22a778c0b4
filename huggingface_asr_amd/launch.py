"""Run one of the reference's scripts UNCHANGED on the HIP classes:

    python -m huggingface_asr_amd.launch [--reference-src DIR] src/trainers/train_enc_dec_asr.py --flag=... (the recipe's own flags)

What it does: puts the reference's `src/` on `sys.path` (default: the script's grand-parent directory, i.e. `<reference>/src` for
`src/trainers/*.py`; or `$HFASR_REFERENCE_SRC`), calls `huggingface_asr_amd.bind.install()` — which rebinds the model classes the reference imports by
name and replaces `utilities.bind.bind_all` — and then executes the script as `__main__` with the remaining arguments.  Under `torchrun` use
`torchrun ... -m huggingface_asr_amd.launch <script> <flags>`; the recipes' `python "$@"` launch lines (`cluster_utilities/LUMI/*.sh`) take
`-m huggingface_asr_amd.launch` in front of the script path."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    src = os.environ.get("HFASR_REFERENCE_SRC")
    if argv and argv[0] == "--reference-src":
        src = argv[1]
        argv = argv[2:]
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    if src is None:
        src = os.path.dirname(os.path.dirname(script))
    if not os.path.isdir(os.path.join(src, "utilities")):
        raise SystemExit(f"{src} does not look like the reference's src/ directory (no utilities/); pass --reference-src")
    for p in (src, os.path.dirname(script)):
        if p not in sys.path:
            sys.path.insert(0, p)
    from huggingface_asr_amd import bind
    bind.install()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
