"""Writes the instantiation lists of the batched external fit's evaluation kernels (vp_inst_extfit_*.hip).
(N basis functions, P pair slots, Q parameters) x capacity 64 R W; per capacity the split R x W is chosen so that the
resident columns leave room for a second wave on the SIMD (fp64: 2 VGPRs per value).  usage: python gen_extfit_inst.py"""

shapes = {
    1: [(1, 1), (2, 2), (3, 3), (4, 2), (4, 4)],
    2: [(1, 1), (2, 2), (3, 3), (4, 2), (4, 4), (6, 3), (6, 4)],
    3: [(2, 2), (3, 3), (4, 2), (4, 4), (6, 3), (6, 4), (6, 6)],
    4: [(3, 3), (4, 4), (6, 3), (6, 4), (6, 6)],
    5: [(4, 4), (6, 6)],
    6: [(5, 5), (6, 6)],
}


def splits(T, n, P, Q):
    # m <= 256, 512, 1024: one wavefront per problem throughout.  Multi-wave groups (R x W = 8 x 2, 4 x 4) were measured for
    # the shapes whose columns fill a wave's registers at R = 16 (fp64, 8 columns): 1.48 / 2.49 ms against 1.19 ms per step of
    # 65 536 problems -- the LDS exchange of every reduction round costs more than the second wave on the SIMD hides.
    return [(4, 1), (8, 1), (16, 1)]


def emit(fname, T, ns, head):
    lines = [head, '#include "vp_extfit.hpp"', '']
    for n in ns:
        for (P, Q) in shapes[n]:
            for (R, W) in splits(T, n, P, Q):
                lines.append('VP_REGISTER_EXTFIT_W(%s, %d, %d, %d, %d, %d)' % (T, n, P, Q, R, W))
    open(fname, 'w').write('\n'.join(lines) + '\n')


def emit_long(fname, T, head):
    # round 5: problems longer than one wave's 1 024 rows -- four waves per problem to 4 096 rows (shapes of up to eight
    # columns).  Eight-wave groups (8 192 rows) were compiled and dropped: the kernel's launch bounds leave them 128 VGPRs
    # per lane (89-746 spilled)
    lines = [head, '#include "vp_extfit.hpp"', '']
    for n in sorted(shapes):
        for (P, Q) in shapes[n]:
            if n + 1 + P <= 8:
                lines.append('VP_REGISTER_EXTFIT_W(%s, %d, %d, %d, %d, %d)' % (T, n, P, Q, 16, 4))
    open(fname, 'w').write('\n'.join(lines) + '\n')


def emit_stream(fname, T, head):
    # end of round 5: any length -- the rows streamed in blocks through the TSQR carry of vp_block.hpp (vp_blk_extfit.hpp); every
    # shape of the resident tables (up to 13 columns: two rows per lane and block, one wave per SIMD)
    lines = [head, '#include "vp_blk_extfit.hpp"', '']
    for n in sorted(shapes):
        for (P, Q) in shapes[n]:
            lines.append('VP_REGISTER_EXTFIT_STREAM(%s, %d, %d, %d)' % (T, n, P, Q))
    open(fname, 'w').write('\n'.join(lines) + '\n')


H = '// batched LM fit of caller-evaluated models (vp_extfit.hpp): evaluation kernels, %s, n = %s: (N, P pair slots, Q parameters, R rows per lane, W waves per problem); written by gen_extfit_inst.py'
emit('vp_inst_extfit_a_f64.hip', 'double', [1, 2], H % ('f64', '1, 2'))
emit('vp_inst_extfit_b_f64.hip', 'double', [3], H % ('f64', '3'))
emit('vp_inst_extfit_c_f64.hip', 'double', [4, 5, 6], H % ('f64', '4, 5, 6'))
emit('vp_inst_extfit_a_f32.hip', 'float', [1, 2, 3], H % ('f32', '1, 2, 3'))
emit('vp_inst_extfit_b_f32.hip', 'float', [4, 5, 6], H % ('f32', '4, 5, 6'))
emit_long('vp_inst_extfit_long_f64.hip', 'double', '// batched LM fit of caller-evaluated models (vp_extfit.hpp): evaluation kernels for LONG problems, f64 (four / eight waves per problem); written by gen_extfit_inst.py')
emit_long('vp_inst_extfit_long_f32.hip', 'float', '// batched LM fit of caller-evaluated models (vp_extfit.hpp): evaluation kernels for LONG problems, f32 (four / eight waves per problem); written by gen_extfit_inst.py')
emit_stream('vp_inst_extfit_stream_f64.hip', 'double', '// batched LM fit of caller-evaluated models at ANY length (vp_blk_extfit.hpp: rows streamed in blocks), f64: (N, P pair slots, Q parameters); written by gen_extfit_inst.py')
emit_stream('vp_inst_extfit_stream_f32.hip', 'float', '// batched LM fit of caller-evaluated models at ANY length (vp_blk_extfit.hpp: rows streamed in blocks), f32: (N, P pair slots, Q parameters); written by gen_extfit_inst.py')
