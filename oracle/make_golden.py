#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerates tests/golden/ from the REFERENCE build.

Run in the build container only (needs oracle/_ref/libspgpu_ref.so, i.e.
/root/reference):    python oracle/make_golden.py

Conversion fixtures: inputs (COO) and the arrays the reference's own converters
(ell.c, hell.c, hdia.cpp compiled unmodified, see oracle/Makefile) produce for
them.  Small cases store the full arrays, the 1024x1024 Laplacian (BASELINE
config 1) stores FNV-1a-64 checksums and the structural numbers.

SpMV fixtures: the reference has no CPU SpMV and no golden values, so expected
z = alpha*A*x + beta*y is computed here from the COO triplets in extended
precision (numpy longdouble, 64-bit mantissa), independent of the oracle's and
the kernels' arithmetic; `scale` = |alpha| * sum_j |a_ij x_j| + |beta y_i| is the
row-wise magnitude the tolerance (1e-6 fp64 / 1e-4 fp32) is taken against.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as O  # noqa: E402
from spgpu_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def exact_spmv(n_rows, rows, cols, vals, base, x, y, alpha, beta):
    """Extended-precision z and the row-wise magnitude scale."""
    cplx = np.iscomplexobj(vals)
    wide = np.clongdouble if cplx else np.longdouble
    r0, c0 = rows.astype(np.int64) - base, cols.astype(np.int64) - base
    prod = vals.astype(wide) * x.astype(wide)[c0]
    acc = np.zeros(n_rows, wide)
    np.add.at(acc, r0, prod)
    mag = np.zeros(n_rows, np.longdouble)
    np.add.at(mag, r0, np.abs(prod))
    z = wide(alpha) * acc + (wide(beta) * y.astype(wide) if beta != 0 else 0)
    scale = abs(alpha) * mag + (np.abs(wide(beta) * y.astype(wide)) if beta != 0 else 0)
    out = np.complex128 if cplx else np.float64
    return z.astype(out), np.asarray(scale, np.float64)


def conversion_case(ref, name, n_rows, n_cols, rows, cols, vals, base, hack_size, with_hdia=True):
    ell = ref.coo_to_ell(n_rows, rows, cols, vals, coo_base=base, ell_base=base)
    hell = ref.ell_to_hell(ell, hack_size)
    out = dict(n_rows=n_rows, n_cols=n_cols, coo_rows=rows, coo_cols=cols, coo_vals=vals, base=base,
               hack_size=hack_size, ell_values=ell["values"], ell_indices=ell["indices"], ell_pitch=ell["pitch"],
               ell_max_row=ell["max_row"], row_lengths=ell["row_lengths"], hell_values=hell["values"],
               hell_indices=hell["indices"], hell_hack_offsets=hell["hack_offsets"], hell_height=hell["height"])
    if with_hdia:
        hdia = ref.coo_to_hdia(n_rows, n_cols, rows, cols, vals, hack_size, coo_base=base)
        out.update(hdia_values=hdia["values"], hdia_offsets=hdia["offsets"], hdia_hack_offsets=hdia["hack_offsets"],
                   hdia_height=hdia["height"])
    return name, out


def add_spmv(case, seed, alpha, beta):
    vals = case["coo_vals"]
    letter = O.LETTER_OF[vals.dtype]
    x = synth.values_for(letter, seed, case["n_cols"])
    y = synth.values_for(letter, seed + 1, case["n_rows"])
    z, scale = exact_spmv(case["n_rows"], case["coo_rows"], case["coo_cols"], vals, case["base"], x, y, alpha, beta)
    case.update(x=x, y=y, alpha=np.asarray(alpha), beta=np.asarray(beta), z_expected=z, z_scale=scale)


def main():
    ref = O.reference_converters()
    if ref is None:
        sys.exit("oracle/_ref/libspgpu_ref.so missing: run `make ref` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    cases = []

    # 1. the reference's ctest.c matrix (single precision, duplicates, rows % 32 != 0)
    n, m, r, c, v = synth.ctest_matrix(np.float32)
    cases.append(conversion_case(ref, "ctest_s", n, m, r, c, v, 0, 32))
    add_spmv(cases[-1][1], 11, 2.0, -3.0)

    # 2. 5-point Laplacian 32x32 and 7-point 16^3 (SURVEY 8(a) known-answer matrices), double
    n, m, r, c, v = synth.laplacian_2d_5pt(32)
    cases.append(conversion_case(ref, "lap2d_32_d", n, m, r, c, v, 0, 32))
    add_spmv(cases[-1][1], 21, 1.0, 0.0)
    n, m, r, c, v = synth.laplacian_3d_7pt(16)
    cases.append(conversion_case(ref, "lap3d_16_d", n, m, r, c, v, 0, 32))
    add_spmv(cases[-1][1], 31, 1.5, 0.5)

    # 3. ragged power-law rows, shuffled COO order, both index bases, hack 32 and 64, all four types
    k = 0
    for letter in "SDCZ":
        for base, hs in ((0, 32), (1, 64)):
            rows_n, cols_n = 500 + 37 * k, 450 + 53 * k
            lengths = synth.power_law_lengths(rows_n, mean=10.0, max_len=96, seed=50 + k)
            lengths[:: 17] = 0  # some empty rows
            n, m, r, c, v = synth.random_rows_coo(rows_n, cols_n, lengths, seed=60 + k, letter=letter, base=base,
                                                  shuffle=True)
            cases.append(conversion_case(ref, f"powerlaw_{letter.lower()}_b{base}_h{hs}", n, m, r, c, v, base, hs))
            alpha = 0.75 if letter in "SD" else complex(0.75, -0.5)
            beta = 0.0 if k % 2 == 0 else (-1.25 if letter in "SD" else complex(-1.25, 0.25))
            add_spmv(cases[-1][1], 70 + k, alpha, beta)
            k += 1

    # 4. degenerate shapes
    e32 = np.zeros(0, np.int32)
    cases.append(conversion_case(ref, "empty_d", 40, 40, e32, e32, np.zeros(0, np.float64), 0, 32))
    cases.append(conversion_case(ref, "onerow_z", 1, 5, np.zeros(3, np.int32), np.array([4, 0, 2], np.int32),
                                 np.array([1 + 2j, -3j, 0.5], np.complex128), 0, 32))

    for name, data in cases:
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)

    # 5. BASELINE config 1 (5-point Laplacian 1024x1024): checksums + structure only
    n, m, r, c, v = synth.laplacian_2d_5pt(1024)
    ell = ref.coo_to_ell(n, r, c, v)
    hell = ref.ell_to_hell(ell, 32)
    hdia = ref.coo_to_hdia(n, m, r, c, v, 32)
    summary = {
        "lap2d_1024_d": dict(
            generator="spgpu_amd.synth.laplacian_2d_5pt(1024), base 0, hackSize 32, outputs zeroed first",
            n=n, nnz=int(r.size), ell_max_row=ell["max_row"], ell_pitch=ell["pitch"], hell_height=hell["height"],
            hell_slots=int(hell["values"].size), hell_last_hack_offset=int(hell["hack_offsets"][-1]),
            hdia_height=hdia["height"],
            fnv=dict(ell_indices=O.fnv(ell["indices"]), ell_values=O.fnv(ell["values"]),
                     hell_indices=O.fnv(hell["indices"]), hell_values=O.fnv(hell["values"]),
                     hell_hack_offsets=O.fnv(hell["hack_offsets"]), row_lengths=O.fnv(ell["row_lengths"]),
                     hdia_offsets=O.fnv(hdia["offsets"]), hdia_hack_offsets=O.fnv(hdia["hack_offsets"]),
                     hdia_values=O.fnv(hdia["values"]))),
        # structural known answers quoted by SURVEY.md 8(a), reproduced by the reference build here
        "survey_8a_structure": dict(
            lap2d_32=dict(n=1024, nnz=4992, ell_max_row=5, ell_pitch=1024, hell_height=158, hell_slots=5056,
                          hell_last_hack_offset=4928),
            lap2d_1024=dict(n=1048576, nnz=5238784, ell_max_row=5, ell_pitch=1048576, hell_height=163776,
                            hell_slots=5240832, hell_last_hack_offset=5240704),
            lap3d_16=dict(n=4096, nnz=27136, hacks=128, hdia_height=880, hack0_offsets=[-16, -1, 0, 1, 16, 256],
                          hack8_offsets=[-256, -16, -1, 0, 1, 16, 256]),
            ctest=dict(ell_max_row=2, ell_pitch=128, hell_height=8, hell_hack_offsets=[0, 64, 128, 192],
                       hdia_hacks=4, hdia_height=4, hdia_hack_offsets=[0, 1, 2, 3, 4])),
    }
    with open(os.path.join(GOLD, "checksums.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", len(cases), "fixtures +", "checksums.json to", GOLD)


if __name__ == "__main__":
    main()
