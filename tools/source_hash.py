"""sha256 over the sources libgpcore.so is built from (gpyreg_amd/csrc/*, include/gpcore.h): the CODE STATE an
evidence file under profiles/ was taken on.  bench.py quotes committed PMC traffic only together with this hash and
says so when it differs from the hash of the sources it is running.  usage: python tools/source_hash.py"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpyreg_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d)) + [os.path.join(ROOT, "include", "gpcore.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
