"""HBM-side traffic per launch of the update path's GEMM-class kernels from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; KiB per dispatch, summed per kernel by tools/pmc_summary.py).

    python tools/pmc_traffic.py fetch.summary.csv write.summary.csv out.json

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-byte requests at 64
bytes, so it is doubled; WRITE_SIZE is taken as is.  bench.py reads the JSON for `roofline.traffic`."""
import csv, json, sys
OPS = [('conv8s4_fwd', 'SwGeom<8, 8, 4, 84, 84, 4,'), ('conv4s2_fwd', 'SwGeom<4, 4, 2, 20, 20, 32,'),
       ('conv3s1_fwd', 'SwGeom<3, 3, 1, 9, 9, 64,'), ('conv3s1_dgrad', 'SwGeom<3, 3, 1, 11, 11, 32,'),
       ('conv3s1_dgrad', 'conv3_dgrad_oa_kernel'),      # overlap-add form (csrc/conv3_dgrad_oa.hpp), the default
       ('conv4s2_dgrad', 'SwGeom<2, 2, 1, 11, 11, 64,'), ('conv8s4_wgrad', 'conv1_wgrad_sa_kernel'),
       ('conv8s4_wgrad', 'conv1_wgrad_ring_kernel'),     # LDS-ring form (csrc/conv_wgrad_sa.hpp), the default for float frames
      
       ('conv4s2_wgrad', 'conv2_wgrad_sa_kernel'), ('conv3s1_wgrad', 'conv3_wgrad_sa_kernel'),
       ('linear1568x512_fwd', 'DenseKCT<true>, acktr::DenseKCT<true>'),
       ('linear1568x512_wgrad', 'fc_wgrad_sa_kernel'),
       ('linear1568x512_dgrad', 'DenseKCT<true>, acktr::DenseRCT<true>'),
       # KFAC input factors (stationary accumulators, csrc/kfac_cov_sa.hpp): bench.py's roofline_kfac operators
       ('kfac_cov_a_conv8s4', 'cov_a_conv1_sa_kernel'), ('kfac_cov_a_conv4s2', 'CovGeom2'), ('kfac_cov_a_conv3s1', 'CovGeom3')]
f = {r['kernel']: r for r in csv.DictReader(open(sys.argv[1]))}
w = {r['kernel']: r for r in csv.DictReader(open(sys.argv[2]))}
out = {}
for op, pat in OPS:
    ks = [k for k in f if pat in k and ('conv_sw_kernel' in k or 'gemm_mfma' in k or 'wgrad_sa_kernel' in k or 'wgrad_ring_kernel' in k or 'cov_a_' in k or 'dgrad_oa_kernel' in k)
          and 'reduce' not in k and 'mirror' not in k]
    if not ks or any(k not in w for k in ks):
        continue
    if len(ks) > 1 and not op.startswith('kfac_'):
        continue
    # (the two channel-group-pair programs of a KFAC factor, DIAG = true / false, are one launch each per update: summed)
    k = ks[0]
    nl = max(int(f[x]['dispatches']) for x in ks)
    fk = sum(float(f[x]['FETCH_SIZE']) for x in ks) / nl
    wk = sum(float(w[x]['WRITE_SIZE']) for x in ks) / nl
    out[op] = dict(kernel=k[:160], launches=nl, fetch_size_kib_raw=round(fk, 1),
                   write_size_kib=round(wk, 1), traffic_bytes=round((2 * fk + wk) * 1024))
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v['traffic_bytes'] for k, v in out.items()}))
