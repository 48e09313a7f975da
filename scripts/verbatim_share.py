"""Share of a file's substantive lines that occur verbatim (whitespace-insensitive) anywhere in the reference's sources - the check VERDICT r3 ran by hand.
usage: python scripts/verbatim_share.py <file> [...]   (needs /root/reference: this container only)"""
import glob, os, re, sys
REF = "/root/reference"
def norm(l): return re.sub(r"\s+", "", l)
def substantive(l):
    t = l.strip()
    return len(norm(t)) >= 12 and not t.startswith(("//", "#", "*", "/*", '"""'))
ref = set()
for pat in ("src/**/*", "include/**/*", "test/**/*", "scripts/**/*"):
    for f in glob.glob(os.path.join(REF, pat), recursive=True):
        if os.path.isfile(f):
            try:
                ref.update(norm(l) for l in open(f, errors="ignore") if substantive(l))
            except OSError:
                pass
for f in sys.argv[1:]:
    lines = [norm(l) for l in open(f, errors="ignore") if substantive(l)]
    hit = sum(1 for l in lines if l in ref)
    print(f"{hit / max(1, len(lines)):.2f}  {hit:4d} / {len(lines):4d}  {f}")
