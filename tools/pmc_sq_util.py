"""profiles/<tag>_pmc_sq*.txt (tools/rocpd_pmc.py table of the SQ pass) -> MFMA utilisation per kernel:
   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x clock x 256 CUs x 4 SIMDs)
(rocprof's derived metric divides by GRBM_GUI_ACTIVE x CU_NUM x 4; GRBM_GUI_ACTIVE was not collected, so the kernel duration
x an assumed 2.0 GHz effective clock under profiling stands in -- MI355X_MICROARCH.md: profiled passes run 1.89-1.95 GHz).
Also the share of wave cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES, both in quad-cycles).
   python tools/pmc_sq_util.py profiles/r3_pmc_sq.txt [pattern ...] > profiles/r3_mfma_util.txt"""
import sys


def table(path):
    rows = {}
    lines = open(path).read().splitlines()
    names = [c.strip() for c in lines[1].split('|')]
    for l in lines[2:]:
        c = [x.strip() for x in l.split('|')]
        rows[c[0]] = dict(zip(names[1:], [float(v) for v in c[1:]]))
    return rows


def main(path, *patterns):
    pats = patterns or ('k_spconv', 'k_rowgemm', 'k_attn', 'k_wgrad', 'k_img_conv3', 'k_img_wgrad9', 'k_rows_wgrad1', 'k_lin_', 'k_expand_bf16', 'k_dconv')
    clock, simds = 2.0e9, 256 * 4
    t = table(path)
    print(f'# MFMA utilisation from {path} (assumed effective clock {clock / 1e9:.1f} GHz, 256 CUs x 4 SIMDs)')
    print('kernel | calls | dur_ms | MfmaUtil | wait share of wave cycles')
    tot_b = tot_t = 0.0
    for k, v in t.items():
        if not any(p in k for p in pats):
            continue
        util = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['dur_ms'] * 1e-3 * clock * simds)
        tot_b += v['SQ_VALU_MFMA_BUSY_CYCLES']
        tot_t += v['dur_ms']
        print(f"{k[:64]} | {int(v['calls'])} | {v['dur_ms']:.2f} | {util:.3f} | {v['SQ_WAIT_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.2f}")
    if tot_t:
        print(f'# family total: {tot_t:.1f} ms, MfmaUtil {tot_b / (tot_t * 1e-3 * clock * simds):.3f}')


if __name__ == '__main__':
    main(*sys.argv[1:])
