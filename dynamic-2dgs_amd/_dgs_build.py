"""In-tree builds of the native libraries, keyed on a HASH of their inputs (sources, headers, compiler flags).

A library is rebuilt when the hash recorded next to it (`<lib>.srchash`) differs from the hash of the current inputs --
file times do not matter, so a checkout, a copy to the GPU box or a touched file can neither force nor hide a rebuild.
`source_hash()` is also what bench.py compares against the hash stored with the PMC traffic figures under profiles/.

Several processes may ask for the same library at once (N ranks of a torchrun launch whose binaries are stale or whose
`.srchash` files did not travel): builds are serialised by an exclusive `flock` on `<lib>.lock`, the compiler writes to a
temporary file in the same directory and the result is moved into place with `os.replace` (atomic on one file system), and
the hash is recorded only after that -- a process that `dlopen`s the path sees either the old complete file or the new
complete file, never a half-written one, and the ranks that waited for the lock find the hash current and reuse the binary.
"""
import fcntl
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


def _current(lib_path, want):
    return os.path.exists(lib_path) and recorded_hash(lib_path) == want


def build(lib_path, cmd, deps, flags, cwd, force=False, verbose=False):
    """Run `cmd` (which must name lib_path as its output) unless lib_path was built from exactly these inputs.
    Returns (lib_path, "compiled" | "reused").  Safe against concurrent callers (module docstring)."""
    want = source_hash(deps, flags)
    if not force and _current(lib_path, want):
        if verbose:
            print("[build] reused   %s (inputs %s)" % (os.path.basename(lib_path), want))
        return lib_path, "reused"
    assert lib_path in cmd, "the build command must name the library as its output"
    with open(lib_path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            # somebody else may have built it while this process waited for the lock
            if not force and _current(lib_path, want):
                if verbose:
                    print("[build] reused   %s (inputs %s, built by another process)" % (os.path.basename(lib_path), want))
                return lib_path, "reused"
            tmp = "%s.tmp%d" % (lib_path, os.getpid())
            tmp_cmd = [tmp if c == lib_path else c for c in cmd]
            if verbose:
                print("[build] compiling %s (inputs %s): %s" % (os.path.basename(lib_path), want, " ".join(cmd)))
            try:
                subprocess.check_call(tmp_cmd, cwd=cwd)
                os.replace(tmp, lib_path)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
            with open(lib_path + ".srchash.tmp%d" % os.getpid(), "w") as f:
                f.write(want + "\n")
            os.replace(lib_path + ".srchash.tmp%d" % os.getpid(), lib_path + ".srchash")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib_path, "compiled"
