"""Synthetic binary genotype-likelihood (GL) inputs, as specified in SURVEY.md §8(d).

Per site an allele frequency q_s ~ U(0.05, 0.5); two haplotype rows per individual form a copying
chain along sites (each haplotype entry is re-drawn ~ Bernoulli(q_s) with probability ``p_resample``,
otherwise copied from the previous site), which gives LD that decays with distance.  Genotype = sum of
the two haplotypes, depth ~ Poisson(depth), alt reads ~ Binomial(depth, [0.01, 0.5, 0.99][g]) and
GL_x = p_x^k (1-p_x)^(d-k): normal scale, un-normalised, laid out ``[site][ind][3]`` as little-endian
doubles -- exactly what the reference's binary reader consumes (shared/read_data.cpp:28-47).

Two generators with the same model: ``numpy`` (fixtures, tests, CPU) and ``torch`` (bench-size
inputs generated directly in HBM).  They use different RNGs, so their streams differ.

Matrices that are NOT SNP-called (the reference's README.md:73 warns about them; its own examples/test.sh feeds
un-called ANGSD output): ``mono_frac`` makes that share of the sites monomorphic in the population (every haplotype
carries the reference allele there, the likelihoods still come from reads with the 1 % error rate; such sites stand outside
the copying chain, so the other sites are the default generator's), ``sfs=True``
draws q_s log-uniform in [0.001, 0.5] instead of U(0.05, 0.5) (a neutral-like site frequency spectrum: at 500
individuals 5-6 % of such sites carry no alternative allele in the sample).  Both default to off: the SURVEY 8(d)
generator is unchanged for the same seed.
"""
from __future__ import annotations

import numpy as np

P_ALT = (0.01, 0.5, 0.99)


def make_positions(n_sites: int, seed: int, max_gap: int = 200, n_chr: int = 1) -> tuple[list[str], np.ndarray]:
    """Chromosome names and 1-based positions; gaps ~ UniformInt[1, max_gap]."""
    rng = np.random.default_rng([seed, 0x706F73])
    gaps = rng.integers(1, max_gap + 1, size=n_sites)
    pos = np.cumsum(gaps)
    chrs = []
    per = -(-n_sites // n_chr)
    for c in range(n_chr):
        lo, hi = c * per, min(n_sites, (c + 1) * per)
        if hi > lo:
            pos[lo:hi] -= pos[lo] - gaps[lo]
            chrs += [f"chr{c + 1}"] * (hi - lo)
    return chrs, pos.astype(np.int64)


def write_pos(path: str, chrs: list[str], pos: np.ndarray, header: bool = False, extra_col: bool = False) -> None:
    with open(path, "w") as fh:
        if header:
            fh.write("chr\tpos\n" if not extra_col else "chr\tpos\tid\n")
        for k, (c, p) in enumerate(zip(chrs, pos)):
            fh.write(f"{c}\t{int(p)}\tsnp{k}\n" if extra_col else f"{c}\t{int(p)}\n")


def make_gl_numpy(n_sites: int, n_ind: int, seed: int, depth: float = 10.0, p_resample: float = 0.05,
                  mono_frac: float = 0.0, sfs: bool = False) -> np.ndarray:
    """Raw (un-normalised, normal-scale) GLs, float64 array [n_sites, n_ind, 3]."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(0.05, 0.5, size=n_sites)
    mono = None
    if sfs or mono_frac > 0:                      # (a stream of its own: the default generator's draws stay as they were)
        rq = np.random.default_rng([seed, 0x6D6F6E6F])
        if sfs:
            q = np.exp(rq.uniform(np.log(0.001), np.log(0.5), size=n_sites))
        mono = rq.random(n_sites) < mono_frac if mono_frac > 0 else None
    n_hap = 2 * n_ind
    fresh = rng.random((n_sites, n_hap)) < q[:, None]
    resample = rng.random((n_sites, n_hap)) < p_resample
    resample[0, :] = True
    src = np.where(resample, np.arange(n_sites)[:, None], 0)
    src = np.maximum.accumulate(src, axis=0)
    hap = np.take_along_axis(fresh, src, axis=0)
    if sfs:
        # the copying chain hands on ALLELES, so a site's sample frequency is mostly its neighbours'; with sfs the chain hands
        # on a latent uniform per haplotype instead and the allele is "latent < q_s": site s has frequency q_s, and LD still
        # decays with distance
        u = np.random.default_rng([seed, 0x6C6174]).random((n_sites, n_hap))
        hap = np.take_along_axis(u, src, axis=0) < q[:, None]
    if mono is not None:                          # monomorphic sites stand OUTSIDE the copying chain: their neighbours keep their LD
        hap = hap & ~mono[:, None]
    g = hap[:, 0::2].astype(np.int64) + hap[:, 1::2].astype(np.int64)
    d = rng.poisson(depth, size=(n_sites, n_ind))
    p_read = np.asarray(P_ALT)[g]
    k = rng.binomial(d, p_read)
    gl = np.empty((n_sites, n_ind, 3), dtype=np.float64)
    for x, px in enumerate(P_ALT):
        gl[:, :, x] = np.power(px, k) * np.power(1.0 - px, d - k)
    return gl


def make_gl_torch(n_sites: int, n_ind: int, seed: int, device, depth: float = 10.0, p_resample: float = 0.05,
                  chunk_sites: int = 16384, mono_frac: float = 0.0, sfs: bool = False):
    """Same model generated on ``device`` with torch; returns a float64 tensor [n_sites, n_ind, 3].

    The copying chain is resolved with a cumulative max over "last re-draw site" per haplotype, done in
    site chunks to bound temporaries (the chain state carried between chunks is one row).
    """
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    gen_q = torch.Generator(device=device)        # mono_frac / sfs: a stream of their own (the default draws stay as they were)
    gen_q.manual_seed(seed ^ 0x6D6F6E6F)
    n_hap = 2 * n_ind
    out = torch.empty((n_sites, n_ind, 3), dtype=torch.float64, device=device)
    p_alt = torch.tensor(P_ALT, dtype=torch.float64, device=device)
    carry = None  # haplotype row of the last site of the previous chunk
    carry_u = None
    for lo in range(0, n_sites, chunk_sites):
        hi = min(n_sites, lo + chunk_sites)
        m = hi - lo
        q = torch.rand((m, 1), generator=gen, device=device, dtype=torch.float64) * 0.45 + 0.05
        if sfs:
            lo_q, hi_q = float(np.log(0.001)), float(np.log(0.5))
            q = torch.exp(torch.rand((m, 1), generator=gen_q, device=device, dtype=torch.float64) * (hi_q - lo_q) + lo_q)
        mono = (torch.rand((m, 1), generator=gen_q, device=device, dtype=torch.float64) < mono_frac) if mono_frac > 0 else None
        fresh = torch.rand((m, n_hap), generator=gen, device=device) < q
        resample = torch.rand((m, n_hap), generator=gen, device=device) < p_resample
        if carry is None:
            resample[0, :] = True
        idx = torch.arange(1, m + 1, device=device, dtype=torch.int32)[:, None]
        src = torch.where(resample, idx, torch.zeros_like(idx))          # 0 = "copy from before the chunk"
        src = torch.cummax(src, dim=0).values.long()
        ext = torch.cat([carry[None, :] if carry is not None else fresh[:1], fresh], dim=0)
        hap = torch.gather(ext, 0, src)
        carry = hap[-1].clone()
        if sfs:                                   # latent uniforms handed on instead of alleles (see make_gl_numpy)
            u = torch.rand((m, n_hap), generator=gen_q, device=device)
            uext = torch.cat([carry_u[None, :] if carry_u is not None else u[:1], u], dim=0)
            ug = torch.gather(uext, 0, src)
            carry_u = ug[-1].clone()
            hap = ug < q.to(ug.dtype)
        if mono is not None:                      # monomorphic sites stand OUTSIDE the copying chain
            hap = hap & ~mono
        g = hap[:, 0::2].long() + hap[:, 1::2].long()
        d = torch.poisson(torch.full((m, n_ind), float(depth), device=device, dtype=torch.float64), generator=gen)
        k = torch.binomial(d, p_alt[g], generator=gen)
        for x in range(3):
            px = p_alt[x]
            out[lo:hi, :, x] = torch.pow(px, k) * torch.pow(1.0 - px, d - k)
        del q, fresh, resample, src, ext, hap, g, d, k
    return out
