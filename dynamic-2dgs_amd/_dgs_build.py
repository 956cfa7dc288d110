"""In-tree builds of the native libraries, keyed on a HASH of their inputs (sources, headers, compiler flags).

A library is rebuilt when the hash recorded next to it (`<lib>.srchash`) differs from the hash of the current inputs --
file times do not matter, so a checkout, a copy to the GPU box or a touched file can neither force nor hide a rebuild.
`source_hash()` is also what bench.py compares against the hash stored with the PMC traffic figures under profiles/.
"""
import hashlib
import os
import subprocess


def source_hash(deps, flags):
    h = hashlib.sha256()
    h.update(("\0".join(flags)).encode())
    for d in deps:
        h.update(b"\0" + os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_hash(lib_path):
    try:
        with open(lib_path + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def build(lib_path, cmd, deps, flags, cwd, force=False, verbose=False):
    """Run `cmd` (which must write lib_path) unless lib_path was built from exactly these inputs.
    Returns (lib_path, "compiled" | "reused")."""
    want = source_hash(deps, flags)
    if not force and os.path.exists(lib_path) and recorded_hash(lib_path) == want:
        if verbose:
            print("[build] reused   %s (inputs %s)" % (os.path.basename(lib_path), want))
        return lib_path, "reused"
    if verbose:
        print("[build] compiling %s (inputs %s): %s" % (os.path.basename(lib_path), want, " ".join(cmd)))
    subprocess.check_call(cmd, cwd=cwd)
    with open(lib_path + ".srchash", "w") as f:
        f.write(want + "\n")
    return lib_path, "compiled"
