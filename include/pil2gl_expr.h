/*
 * pil2gl_expr.h -- binary encoding of the reference's expression op-list.
 *
 * The reference evaluates constraint / FRI / intermediate polynomials by turning
 * `code.code` (an array of {op, dest, src[]}; src/pil_info/helpers/code/codegen.js:75-125,257-283)
 * into JavaScript source and calling it once per row
 * (src/prover/prover_helpers.js:31-45 calculateExps, :83-107 compileCode,
 *  :109-150 setRef, :152-218 getRef, :220-259 evalMap).
 * This header is the C form of that op-list.  The JS-side encoder
 * (pil2-stark-js_amd/js/prover_helpers.js) maps every reference operand kind
 * onto one of three operand classes:
 *
 *   reference ref.type                         -> encoding
 *   -----------------------------------------------------------------------------
 *   tmp{id,dim}                                -> GLX_TMP  id
 *   const{id,prime}        ctx.const_n/_ext    -> GLX_SEC  section=const, offset=id, prime
 *   cm{id,prime,dim}       evalMap(): stage buffer cm{s}_n/_ext, stagePos
 *                                              -> GLX_SEC  section=cm{s}, offset=stagePos, prime
 *   x                      ctx.x_n / x_ext     -> GLX_SEC  section=x (width 1)
 *   Zi{boundaryId}         ctx.Zi_ext[zi*extN+i] -> GLX_SEC section=Zi#zi (width 1)
 *   xDivXSubXi{id}         [3*(id + nOpen*i)..] -> GLX_SEC section=xDivXSubXi (width 3*nOpen), offset=3*id
 *   q / f (destinations)   q_ext / f_ext       -> GLX_SEC  section=q|f, offset 0
 *   number / public / challenge / subproofValue / eval
 *                                              -> GLX_SCALAR index into ctx->scalars (u64 words)
 *
 * Row addressing follows evalMap (prover_helpers.js:220-233): an operand with
 * row offset `prime` on a domain of 2^nBits rows is read at row
 * (i + prime * 2^primeShift) mod 2^nBits, primeShift = 0 on domain "n" and
 * nBitsExt-nBits on domain "ext".
 *
 * Arithmetic is F3g.add/sub/mul on mixed dim-1/dim-3 operands
 * (src/helpers/f3g.js:47-104), including sub(scalar, triple) negating
 * components 1 and 2 (f3g.js:66).
 */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { GLX_OP_ADD = 0, GLX_OP_SUB = 1, GLX_OP_MUL = 2, GLX_OP_COPY = 3 };
enum { GLX_TMP = 0, GLX_SEC = 1, GLX_SCALAR = 2 };

typedef struct {
    uint8_t  kind;      /* GLX_TMP | GLX_SEC | GLX_SCALAR */
    uint8_t  dim;       /* 1 or 3 */
    uint16_t section;   /* GLX_SEC: index into glx_ctx.sections */
    int32_t  prime;     /* GLX_SEC: row offset in base-domain rows */
    uint32_t index;     /* GLX_TMP: tmp slot; GLX_SEC: column offset; GLX_SCALAR: word offset */
    uint32_t pad_;
} glx_ref;              /* 16 bytes */

typedef struct {
    uint32_t op;        /* GLX_OP_* */
    uint32_t pad_;
    glx_ref  dest;
    glx_ref  src[2];    /* src[1] unused for COPY */
} glx_op;               /* 56 bytes */

typedef struct {
    uint64_t *ptr;      /* row-major rows x width, canonical u64 */
    uint64_t  width;
} glx_section;

typedef struct {
    uint32_t nBits;         /* log2(rows) of the evaluated domain */
    uint32_t primeShift;    /* 0 ("n") or nBitsExt-nBits ("ext") */
    uint32_t nSections;
    uint32_t nScalars;      /* u64 words in scalars[] */
    const glx_section *sections;
    const uint64_t    *scalars;
} glx_ctx;

typedef struct {
    uint32_t nOps;
    uint32_t nTmp;          /* code.tmpUsed: number of tmp slots (each holds up to 3 u64) */
    const glx_op *ops;
} glx_program;

#ifdef __cplusplus
}
#endif
