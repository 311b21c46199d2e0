"""Synthetic sequences for benchmarks and parity tests.

Every model is a pure function (model, seed, n, position) -> encoded symbol
(0..sigma-1 letters, 254 wildcard, 255 separator), built from the splitmix64
output function only, so the numpy code here and the HIP kernel
``gtamd_synth_bytes`` (genometools_amd/csrc/esa_synth.hip) produce identical
bytes and a CPU oracle can be fed the very same sequence at any size.

Models:
  MODEL_UNIFORM_DNA (0)  i.i.d. uniform ACGT, one sequence, no specials
                         (BASELINE.json configs[1]).
  MODEL_HUMANLIKE_DNA (1) "human-like" DNA (configs[2]): uniform background;
                         10 % of the 8192-base blocks are copies of other
                         blocks (1/8 of them of 16 high-copy family blocks)
                         with 0 / 0.1 / 1 / 5 % point mutations, so the LCP
                         distribution has a heavy tail and .llv is not empty;
                         0.1 % of the blocks carry a tandem repeat (period 1-6,
                         64-2111 bases); about 2 % N in runs of 20-5200 bases;
                         isolated IUPAC wildcards at rate 2^-18; 24 sequences
                         (23 separators) with the length proportions of the
                         human chromosomes when n >= 65536.
  MODEL_REPEAT_HEAVY (3) the hard case for prefix doubling: the human-like model
                         with HALF of the 8192-base blocks copies of other
                         blocks (same family share and mutation levels) and up
                         to four satellite arrays -- exact tandem repeats of
                         period 171 / 5 / 42 / 68 over 100 000 to 1 000 000
                         bases (capped at n/40) -- so that LCP values reach
                         10^5..10^6 and the refinement needs 14-16 rounds.
  MODEL_PROTEIN (2)      i.i.d. residues with Swiss-Prot-like frequencies over
                         LVIFKREDAGSTNQYWPHMC (codes 0..19), sequences of mean
                         length ~330 (separator probability 1/331, never two
                         in a row), wildcard X at rate 2^-13 (configs[4]).
"""
import numpy as np

MODEL_UNIFORM_DNA = 0
MODEL_HUMANLIKE_DNA = 1
MODEL_PROTEIN = 2
MODEL_REPEAT_HEAVY = 3

WILDCARD = 254
SEPARATOR = 255

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

BLK_SHIFT = 13
BLK = 1 << BLK_SHIFT
DUP_T = 6554
DUP_T_HEAVY = 32768
SAT_LEN = (100000, 250000, 500000, 1000000)
SAT_PER = (171, 5, 42, 68)
SAT_BASE = 1 << 40
TANDEM_T = 64
NRUN_T = 10748
MUT_THR = (0, 4294967, 42949673, 214748365)
NRUN_LEN = (20, 50, 90, 140, 200, 280, 370, 480, 620, 800, 1000, 1300, 1700,
            2300, 3300, 5200)
CHROM_CUM = (5268, 10409, 14615, 18652, 22518, 26151, 29528, 32609, 35540,
             38387, 41255, 44080, 46502, 48775, 50942, 52854, 54617, 56316,
             57570, 58929, 59928, 61011, 64325)
PROT_CUM = (6340, 10848, 14740, 17273, 21106, 24735, 29165, 32742, 38156,
            42796, 47101, 50606, 53270, 55849, 57766, 58474, 61559, 63049,
            64637, 65536)
PROT_SEP_T = 198      # of 65536: 1/331
PROT_X_MASK = 0x1FFF  # wildcard when (h >> 20) & mask == 0


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    with np.errstate(over="ignore"):
        return _mix64(np.uint64(seed) + np.uint64(stream) * _GOLD
                      + np.uint64(0x1234567))


def _h(key, x):
    with np.errstate(over="ignore"):
        return _mix64(key + (np.asarray(x, dtype=np.uint64) + np.uint64(1)) * _GOLD)


def _bg(k0, q):
    q = np.asarray(q, dtype=np.uint64)
    w = _h(k0, q >> np.uint64(5))
    return ((w >> (np.uint64(2) * (q & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)


def separators(n):
    """positions of the 23 separators of the human-like models"""
    if n < 65536:
        return np.zeros(0, dtype=np.uint64)
    return np.array([(n * c) >> 16 for c in CHROM_CUM], dtype=np.uint64)


def _uniform(seed, n, lo, hi):
    return _bg(_stream_key(seed, 0), np.arange(lo, hi, dtype=np.uint64))


def satellites(n):
    """(start, length, period) of the satellite arrays of MODEL_REPEAT_HEAVY"""
    nsat = 4 if n >= (1 << 24) else (2 if n >= (1 << 16) else 0)
    return [((n * (2 * k + 1)) // 9, min(n // 40, SAT_LEN[k]), SAT_PER[k]) for k in range(nsat)]


def _repeatheavy(seed, n, lo, hi):
    return _humanlike(seed, n, lo, hi, DUP_T_HEAVY, satellites(n))


def _humanlike(seed, n, lo, hi, dup_t=DUP_T, sats=()):
    p = np.arange(lo, hi, dtype=np.uint64)
    k = [_stream_key(seed, s) for s in range(6)]
    nblocks = np.uint64((n + BLK - 1) >> BLK_SHIFT)
    b = p >> np.uint64(BLK_SHIFT)
    o = p & np.uint64(BLK - 1)
    # per block parameters (computed per position for simplicity: the block
    # hashes are cheap next to the per-position ones)
    ub, inv = np.unique(b, return_inverse=True)
    kind = (_h(k[1], ub) & np.uint64(0xFFFF)).astype(np.int64)
    hb2 = _h(k[2], ub)
    hn = _h(k[4], ub)
    is_dup = (kind < dup_t)[inv]
    is_tan = ((kind >= dup_t) & (kind < dup_t + TANDEM_T))[inv]
    hb2p = hb2[inv]
    out = _bg(k[0], p)
    # duplicated blocks
    if is_dup.any():
        d = np.nonzero(is_dup)[0]
        h2 = hb2p[d]
        family = (h2 & np.uint64(7)) == 0
        src = np.where(family, (h2 >> np.uint64(8)) & np.uint64(15),
                       (h2 >> np.uint64(8)) % nblocks)
        thr = np.array(MUT_THR, dtype=np.uint64)[((h2 >> np.uint64(3)) & np.uint64(3)).astype(np.int64)]
        q = (src << np.uint64(BLK_SHIFT)) | o[d]
        c = _bg(k[0], q)
        hm = _h(k[3], p[d])
        mut = (hm & np.uint64(0xFFFFFFFF)) < thr
        c = np.where(mut, (c + 1 + ((hm >> np.uint64(32)) % np.uint64(3)).astype(np.uint8)) & 3, c)
        out[d] = c.astype(np.uint8)
    # tandem repeats
    if is_tan.any():
        d = np.nonzero(is_tan)[0]
        h2 = hb2p[d]
        per = np.uint64(1) + ((h2 >> np.uint64(40)) % np.uint64(6))
        so = (h2 >> np.uint64(8)) & np.uint64(4095)
        tl = np.uint64(64) + ((h2 >> np.uint64(20)) & np.uint64(2047))
        od = o[d]
        inside = (od >= so) & (od < so + tl)
        q = (b[d] << np.uint64(BLK_SHIFT)) + so + ((od - so) % per)
        c = _bg(k[0], np.where(inside, q, p[d]))
        out[d] = c
    # satellite arrays (exact tandem repeats over 10^5..10^6 bases)
    for si, (st, ln, per) in enumerate(sats):
        a, b = max(lo, st), min(hi, st + ln)
        if a < b:
            q = np.uint64(SAT_BASE + 4096 * si) + \
                (np.arange(a, b, dtype=np.uint64) - np.uint64(st)) % np.uint64(per)
            out[a - lo:b - lo] = _bg(k[0], q)
    # N runs
    hnp = hn[inv]
    has = (hnp & np.uint64(0xFFFF)) < np.uint64(NRUN_T)
    ns = (hnp >> np.uint64(16)) & np.uint64(BLK - 1)
    nl = np.array(NRUN_LEN, dtype=np.uint64)[((hnp >> np.uint64(32)) & np.uint64(15)).astype(np.int64)]
    inrun = has & (o >= ns) & (o < ns + nl)
    out[inrun] = WILDCARD
    # isolated IUPAC wildcards
    out[(_h(k[5], p) & np.uint64(0x3FFFF)) == 0] = WILDCARD
    # separators
    for s in separators(n):
        if lo <= s < hi:
            out[int(s) - lo] = SEPARATOR
    return out


def _protein(seed, n, lo, hi):
    p = np.arange(lo, hi, dtype=np.uint64)
    k0 = _stream_key(seed, 0)
    h = _h(k0, p)
    r = (h & np.uint64(0xFFFF)).astype(np.int64)
    out = np.searchsorted(np.array(PROT_CUM, dtype=np.int64), r, side="right").astype(np.uint8)
    out[((h >> np.uint64(20)) & np.uint64(PROT_X_MASK)) == 0] = WILDCARD

    def raw(q):
        hq = _h(k0, q)
        return (((hq >> np.uint64(40)) & np.uint64(0xFFFF)) < np.uint64(PROT_SEP_T)) \
            & (q > 0) & (q + np.uint64(1) < np.uint64(n))
    sep = raw(p) & ~raw(p - np.uint64(1))
    out[sep] = SEPARATOR
    return out


def generate(model, seed, n, lo=0, hi=None):
    """encoded symbols [lo, hi) of the synthetic sequence of length n"""
    hi = n if hi is None else hi
    if hi <= lo:
        return np.zeros(0, dtype=np.uint8)
    fn = {MODEL_UNIFORM_DNA: _uniform, MODEL_HUMANLIKE_DNA: _humanlike,
          MODEL_PROTEIN: _protein, MODEL_REPEAT_HEAVY: _repeatheavy}[model]
    chunks = []
    step = 1 << 22
    for a in range(lo, hi, step):
        chunks.append(fn(seed, n, a, min(hi, a + step)))
    return np.concatenate(chunks)


def numofchars(model):
    return 20 if model == MODEL_PROTEIN else 4


DNA_LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)
PROTEIN_LETTERS = np.frombuffer(b"LVIFKREDAGSTNQYWPHMC", dtype=np.uint8)


def write_fasta(path, enc, protein=False, width=70):
    """write encoded symbols as (multi-)FASTA that the reference's encoder maps
    back to exactly `enc` (wildcards as N / X, separators as record breaks)"""
    enc = np.asarray(enc, dtype=np.uint8)
    table = np.zeros(256, dtype=np.uint8)
    letters = PROTEIN_LETTERS if protein else DNA_LETTERS
    table[:len(letters)] = letters
    table[WILDCARD] = ord("X") if protein else ord("N")
    cuts = np.nonzero(enc == SEPARATOR)[0]
    starts = np.concatenate(([0], cuts + 1))
    ends = np.concatenate((cuts, [len(enc)]))
    with open(path, "wb") as f:
        for i, (a, b) in enumerate(zip(starts, ends)):
            f.write(b">synth%d\n" % i)
            txt = table[enc[a:b]]
            full = (len(txt) // width) * width
            if full:
                rows = txt[:full].reshape(-1, width)
                block = np.empty((rows.shape[0], width + 1), dtype=np.uint8)
                block[:, :width] = rows
                block[:, width] = 10
                f.write(block.tobytes())
            if full < len(txt):
                f.write(txt[full:].tobytes() + b"\n")
