"""Dev tool (CPU): differential fuzz of the HOST readers against the reference's own compiled readers (oracle/_ref).

Small valid files -- a text genotype file (called genotypes or likelihood triples), a positions file -- are mutated a few times
(lines deleted / doubled / emptied / turned into words, fields dropped or replaced by words, out-of-range or fractional numbers,
separators changed, CR added, final newline removed, label columns, comment lines ...) and read three times: by the reference's
read_geno / read_dist (in a forked child: its error() ends the process, and one of its loops never ends on a header line
further down, read_data.cpp:188-195), by the oracle's restatement and by the product's ngsld_host_read_geno_text /
ngsld_host_read_pos.  Where the reference returns values, both others must return the same values; where it ends in an error,
both must fail with its message.  Known, documented exceptions are counted, not hidden.
python tools/reader_fuzz.py [geno|pos] [first] [last]"""
import ctypes as C
import math
import os
import re
import signal
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from ngsld_amd import capi  # noqa: E402
from oracle import orc  # noqa: E402

WORDS = ["NA", "x", "marker", "3", "-2", "1.5", "0.4", "1e0", "2.", "nan", "inf", "-1", "", "0x1", "1 ", "+1"]


def mutate(lines, rng, n_mut, sep_ok=True):
    lines = list(lines)
    for _ in range(n_mut):
        kind = int(rng.integers(0, 12))
        k = int(rng.integers(0, max(1, len(lines))))
        if not lines:
            break
        if kind == 0:
            del lines[k]
        elif kind == 1:
            lines.insert(k, lines[k])
        elif kind == 2:
            lines.insert(k, "")
        elif kind == 3:
            lines.insert(k, "\t".join(rng.choice(["marker", "ref", "alt", "id", "pos", "7"], size=int(rng.integers(1, 4)))))
        elif kind == 4:
            f = lines[k].split("\t")
            lines[k] = "\t".join(f[:int(rng.integers(0, len(f) + 1))])
        elif kind == 5:
            f = lines[k].split("\t")
            f[int(rng.integers(0, len(f)))] = str(rng.choice(WORDS))
            lines[k] = "\t".join(f)
        elif kind == 6 and sep_ok:
            lines[k] = lines[k].replace("\t", str(rng.choice([" ", "  ", "\t\t", " \t"])))
        elif kind == 7:
            lines[k] += str(rng.choice(["\r", "\t", " ", "\textra"]))
        elif kind == 8:
            lines[k] = str(rng.choice(["lab\t", "chrZ_1\tA\tC\t", "9\t"])) + lines[k]
        elif kind == 9:
            lines.insert(k, "#" + lines[k])
        elif kind == 10:
            lines = lines[:k]
        elif kind == 11:
            lines[k] = lines[k].upper() if rng.random() < 0.5 else lines[k] + lines[k]
    return lines


def in_child(fn, timeout=3):
    """fn() in a forked child: (exit status or None when it had to be killed, its stderr)."""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        os.dup2(w, 2)
        signal.alarm(timeout)
        try:
            fn()
            os._exit(0)
        except BaseException:
            os._exit(99)
    os.close(w)
    chunks = []
    while True:
        b = os.read(r, 65536)
        if not b:
            break
        chunks.append(b)
        if sum(map(len, chunks)) > 1 << 20:   # (the endless loop prints a line per turn)
            break
    os.close(r)
    _, st = os.waitpid(pid, 0) if sum(map(len, chunks)) <= 1 << 20 else (os.kill(pid, signal.SIGKILL), os.waitpid(pid, 0)[1])
    code = os.WEXITSTATUS(st) if os.WIFEXITED(st) else None
    return code, b"".join(chunks).decode(errors="replace")


def ref_message(stderr):
    m = re.search(r"ERROR: \[(\w+)\] (.*)", stderr)
    return (m.group(1), m.group(2)) if m else ("", "")


def geno_case(k, d):
    rng = np.random.default_rng(910_000 + k)
    n_ind, n_sites = int(rng.integers(1, 5)), int(rng.integers(1, 7))
    probs = bool(rng.random() < 0.4)
    if probs:
        v = rng.dirichlet([1, 1, 1], size=(n_sites, n_ind))
        rows = ["\t".join(repr(float(x)) for x in r.reshape(-1)) for r in v]
    else:
        rows = ["\t".join(str(int(x)) for x in r) for r in rng.integers(-1, 3, size=(n_sites, n_ind))]
    if rng.random() < 0.3:
        rows = ["marker\tallele"] + rows
    lines = mutate(rows, rng, int(rng.integers(0, 4)))
    text = "\n".join(lines) + ("\n" if rng.random() < 0.85 and lines else "")
    path = os.path.join(d, f"g{k}.geno")
    with open(path, "w", newline="") as fh:
        fh.write(text)
    R = orc.ref()
    out = np.full((n_sites, n_ind, 3), np.nan)
    shared = os.path.join(d, f"g{k}.npy")

    def child():
        R.ref_read_geno_text(path.encode(), int(probs), 0, n_ind, n_sites, orc.dp(out))
        np.save(shared, out)
    code, stderr = in_child(child)
    gl_orc = np.empty((n_sites, n_ind, 3))
    err = C.create_string_buffer(256)
    rc_orc = orc.lib().orc_read_geno_text(path.encode(), int(probs), 0, n_ind, n_sites, orc.dp(gl_orc), err, 256)
    try:
        raw, is_log = capi.read_geno_text(path, probs, False, n_ind, n_sites)
        msg_hip = None
    except capi.NgsldError as e:
        raw, msg_hip = None, e.msg
    note = None
    if code == 0:
        want = np.load(shared)
        if raw is None or rc_orc != 0:
            if "empty line" in (msg_hip or "") and "empty line" in err.value.decode():
                return "documented: empty line in a site's place", None
            return None, f"reference returned values; oracle rc {rc_orc} ({err.value.decode()}), product {msg_hip!r}"
        got = raw.copy()
        with np.errstate(all="ignore"):
            for t in got.reshape(-1, 3):
                if not is_log:   # (libm's log, as the reference calls it: numpy's vectorised log rounds differently now and then)
                    t[:] = [math.log(x) if x > 0 else (-math.inf if x == 0 else math.nan) for x in t]
                orc.lib().orc_post_prob(orc.dp(t), orc.dp(t.copy()), 3)
        if not np.array_equal(gl_orc, want, equal_nan=True):
            return None, "oracle's values differ from the reference's"
        if not np.array_equal(got, want, equal_nan=True):
            return None, "product's values differ from the reference's"
    elif code is None:
        note = "reference did not end"
        if raw is not None or rc_orc == 0:
            return None, "reference did not end, but a reader returned values"
    else:
        fn, msg = ref_message(stderr)
        if raw is not None or rc_orc == 0:
            return None, f"reference failed ({msg}); oracle rc {rc_orc}, product {'values' if raw is not None else msg_hip!r}"
        if msg_hip != err.value.decode():
            return None, f"product {msg_hip!r} against the oracle's {err.value.decode()!r}"
        if msg_hip != msg:
            return None, f"messages: reference {msg!r}, product {msg_hip!r}"
    return note or "same", None


def pos_case(k, d):
    rng = np.random.default_rng(920_000 + k)
    n_sites = int(rng.integers(1, 8))
    n_chr = int(rng.integers(1, 3))
    pos = np.cumsum(rng.integers(1, 500, size=n_sites))
    rows = [f"chr{1 + (s * n_chr) // n_sites}\t{int(p)}" + ("\tsnp%d" % s if k % 3 == 0 else "") for s, p in enumerate(pos)]
    header = bool(rng.random() < 0.3)
    if header:
        rows = ["chr\tpos" + ("\tid" if k % 3 == 0 else "")] + rows
    lines = mutate(rows, rng, int(rng.integers(0, 4)), sep_ok=False)
    # (more lines than n_sites overflow the reference's own array before it counts them, read_data.cpp:172-179: not comparable)
    usable = [ln for ln in lines if ln and not ln.startswith("#")]
    if len(usable) - (1 if header else 0) > n_sites:
        return "skipped: more lines than sites (the reference writes past its array)", None
    text = "\n".join(lines) + ("\n" if lines else "")
    path = os.path.join(d, f"p{k}.pos")
    with open(path, "w", newline="") as fh:
        fh.write(text)
    R = orc.ref()
    out = np.full(n_sites, np.nan)
    shared = os.path.join(d, f"p{k}.npy")

    def child():
        R.ref_read_dist(path.encode(), int(header), n_sites, orc.dp(out))
        np.save(shared, out)
    code, stderr = in_child(child)
    L = capi.lib()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = L.ngsld_host_read_pos(path.encode(), int(header), n_sites, C.byref(h), err, 256)
    got = None
    if rc == 0:
        L.ngsld_host_pos_dist.restype = C.POINTER(C.c_double)
        got = np.ctypeslib.as_array(L.ngsld_host_pos_dist(h), shape=(n_sites,)).copy()
        L.ngsld_host_free_pos(h)
    msg_hip = err.value.decode()
    if code == 0:
        want = np.load(shared)
        if got is None:
            if "newline" in msg_hip:
                return "documented", None
            return None, f"reference returned distances; product {msg_hip!r}"
        if not np.array_equal(got, want, equal_nan=True):
            return None, f"distances differ: {got} against {want}"
    elif code is None:
        if got is not None:
            return None, "reference did not end, product returned distances"
        return "reference did not end (header line further down); product: " + msg_hip, None
    else:
        fn, msg = ref_message(stderr)
        if got is not None:
            # a last line without newline is lost to the reference (gen_func.cpp:253), kept here
            if not text.endswith("\n"):
                return "documented: last line without newline kept", None
            return None, f"reference failed ({msg}); product returned {got}"
        if msg_hip != msg:
            if not text.endswith("\n"):
                return "documented: last line without newline kept", None
            return None, f"messages: reference [{fn}] {msg!r}, product {msg_hip!r}"
    return "same", None


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "geno"
    first, last = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 2000)
    if orc.ref() is None:
        raise SystemExit("oracle/_ref is not built (oracle/build_ref.sh)")
    tally, bad = {}, 0
    with tempfile.TemporaryDirectory() as d:
        for k in range(first, last):
            note, diff = (geno_case if what == "geno" else pos_case)(k, d)
            if diff is not None:
                bad += 1
                print(f"case {k}: {diff}", flush=True)
            else:
                key = note.split(";")[0] if note.startswith("reference did not end") else note
                tally[key] = tally.get(key, 0) + 1
    print(f"reader fuzz ({what}): cases {first}..{last - 1}, {bad} differ; " + ", ".join(f"{v} {k}" for k, v in sorted(tally.items())))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
