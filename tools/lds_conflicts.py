"""Enumerate LDS bank conflicts of the MFMA fragment reads (ds_read_b128) for a tile layout.

MI355X_MICROARCH.md (LDS table): a wave64 ds_read_b128 is serviced in four NON-contiguous 16-lane groups, one LDS cycle each
when the 16 pieces of a group fall on 16 distinct 16-byte slots of the 256-byte bank row; every extra distinct address on a
slot adds a cycle.  A fragment read of the conv kernels: lane = (li = lane & 15, kq = lane >> 4) reads 16 bytes at
row (base + li), k-granule kq (+ 4 * h for the second half of a 64-channel chunk), swizzled by a per-row key.

  python tools/lds_conflicts.py        # prints the worst multiplicity per layout (1 = conflict-free)
"""
GROUPS = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)],
          [*range(32, 36), *range(44, 48), *range(52, 60)], [*range(36, 44), *range(48, 52), *range(60, 64)]]
CONTIG = [list(range(i * 16, i * 16 + 16)) for i in range(4)]


def worst(row_bytes, key, halves=(0,), groups=GROUPS):
    w = 0
    for h in halves:
        for g in groups:
            slots = {}
            for lane in g:
                li, kq = lane & 15, lane >> 4
                addr = li * row_bytes + (((h * 4 + kq) ^ key(li)) * 16)
                slots.setdefault((addr // 16) % 16, set()).add(addr)
            w = max(w, max(len(v) for v in slots.values()))
    return w


LAYOUTS = {
    'round-2 ping-pong tiles: 64-B rows, key (row >> 2) & 3': (64, lambda r: (r >> 2) & 3, (0,)),
    'round-3 tiles: 64-B rows, key (3 * (row >> 2)) & 3': (64, lambda r: ((r >> 2) * 3) & 3, (0,)),
    'LDS-DMA tiles, 64-channel chunks: 128-B rows, key (row >> 1) & 7': (128, lambda r: (r >> 1) & 7, (0, 1)),
    'padded 80-B rows, no key (weight-gradient tiles)': (80, lambda r: 0, (0,)),
    'unpadded 64-B rows, no key': (64, lambda r: 0, (0,)),
}

def worst_tr():
    """transposed reads (ds_read_b64_tr_b16: two 32-lane groups, bank = (addr / 4) % 64, 8 bytes per lane) of the experimental
    weight-gradient tile k_spconv_wgrad_bf16_tr: 256-byte pair rows, granule g of pair p at slot g ^ key(p)"""
    key = lambda p: ((p & 3) | (((p >> 3) & 1) << 2)) << 1
    w = 0
    for cb in range(8):
        for r in range(2):
            for grp in (range(0, 32), range(32, 64)):
                banks = {}
                for lane in grp:
                    li, kq = lane & 15, lane >> 4
                    pr = kq * 8 + r * 4 + (li >> 2)
                    g = cb * 2 + ((li & 3) >> 1)
                    a = pr * 256 + ((g ^ key(pr)) * 16) + (li & 1) * 8
                    for b in ((a // 4) % 64, (a // 4 + 1) % 64):
                        banks.setdefault(b, set()).add(a)
                w = max(w, max(len(v) for v in banks.values()))
    return w


if __name__ == '__main__':
    for name, (rb, key, hs) in LAYOUTS.items():
        print(f'{name}: documented lane groups {worst(rb, key, hs)}-way, contiguous groups {worst(rb, key, hs, CONTIG)}-way')
    print(f'transposed fragment reads of the experimental weight-gradient tile (ds_read_b64_tr_b16): {worst_tr()}-way')
    for rb in range(64, 177, 16):
        print(f'padded rows of {rb} B, no key: {worst(rb, lambda r: 0)}-way (documented groups)')
