"""Identity of the source tree a profile was taken on: sha256 over the library sources, the package, bench.py and the
header, printable on the GPU box (which has no .git).  tools/collect_profiles.sh stores it beside the raw profiles; the
summarising tools stamp their output with `git describe` of the commit whose tree has THIS id (or say that none has)."""
import glob, hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ("cgat_amd/csrc/*.hip", "cgat_amd/csrc/*.h", "cgat_amd/*.py", "cgat_amd/build_lib.sh", "include/*.h", "bench.py")


def tree_id(root=ROOT):
    h = hashlib.sha256()
    for pat in PATTERNS:
        for f in sorted(glob.glob(os.path.join(root, pat))):
            h.update(os.path.relpath(f, root).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stamp(collection_dir=None):
    """'<git describe> (source tree <id>)' for the working tree, checked against the id the collection recorded."""
    cur = tree_id()
    try:
        desc = subprocess.check_output(["git", "-C", ROOT, "describe", "--always"], text=True).strip()
        # dirty = uncommitted changes to the SOURCES the id covers (profiles / docs written while publishing do not count)
        dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--"] + list(PATTERNS), text=True).strip()
        if dirty:
            desc += "-dirty"
    except Exception:
        desc = "no-git"
    s = f"{desc} (source tree {cur}"
    rec = None
    if collection_dir and os.path.exists(os.path.join(collection_dir, "tree_id.txt")):
        rec = open(os.path.join(collection_dir, "tree_id.txt")).read().strip()
        s += "; the profiled build's own tree id " + (rec + (": the same tree" if rec == cur else ": A DIFFERENT TREE"))
    return s + ")"


if __name__ == "__main__":
    print(tree_id() if len(sys.argv) < 2 else stamp(sys.argv[1]))
