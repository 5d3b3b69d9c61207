// fc_tuning.h -- the switches of TUNING BUILDS (make BUILD=build_x OUT=../libfc_hip_x.so EXTRA="-DFC_TUNING_BUILD -DFC_...",
// selected at run time with FC_LIB_PATH), in one place.  None of them is defined in the product build; several of them
// produce WRONG results on purpose (ablations: a stage of a kernel left out to see what it costs) or add in-kernel time
// stamps.  A translation unit that sees one of them without FC_TUNING_BUILD refuses to compile, so that no such switch
// can slip into libfc_hip.so through an EXTRA left over in a shell.
//
//   timelines (s_memtime stamps into a side buffer; results unchanged, timing not):   FC_TIMELINE  FC_H2_TIMELINE  FC_RB_TIMELINE
//     FC_TFD_STAMPS (cfg3's three steps: fc_tfd_core.h -- cycles per phase of the largest TFD component, tools/ladder_stamps.py;
//       fc_prune.hip -- phases of the first-match walk per workgroup, tools/fm_stamps.py; fc_torsion.hip -- per-node phases of the
//       scan tree's last level, tools/ts_stamps.py; with FC_TFD_STAMPS_SMALL the components of 19 ... 76 nodes instead of the
//       largest, tools/comp_stamps_small.py; phases of chunk_front, tools/cf_stamps.py)
//   ablations -- WRONG RESULTS (a phase of the kernel skipped):
//     split-half / fp32 screens:  FC_H2_ABLATE_K  FC_H2_ABLATE_ROWS  FC_H2_ABLATE_POLY  FC_F32_ABLATE_K  FC_F32_ABLATE_POLY
//     candidate staging:          FC_ABLATE_PUSH  FC_ABLATE_STAGE  FC_ABLATE_OVERFLOW  FC_ABLATE_REDO
//     bucket refine:              FC_RB_NOSTAGE  FC_RB_NOPASS1  FC_RB_NOPASS2  FC_RB_NOJACOBI  FC_RB_NOCOMPUTE  FC_RB_ROWS_FROM_LDS
//                                 FC_RB_SAMEROW
//   shapes (results unchanged):   FC_RB_ROWS  FC_RB_COLSHIFT  FC_RB_WPS  FC_RB_AHEAD  FC_RB_CHUNK  FC_H2_WGS  FC_F32_WGS
//                                 FC_REFINE_ROUNDS  FC_V2_ALIGN  FC_STAGE_PAIRS_F32  FC_TS_WGS  FC_TS_TURN
#pragma once

#if !defined(FC_TUNING_BUILD) &&                                                                                             \
    (defined(FC_TIMELINE) || defined(FC_H2_TIMELINE) || defined(FC_RB_TIMELINE) || defined(FC_TFD_STAMPS) || defined(FC_TFD_STAMPS_SMALL) || defined(FC_H2_ABLATE_K) ||   \
     defined(FC_H2_ABLATE_ROWS) || defined(FC_H2_ABLATE_POLY) || defined(FC_F32_ABLATE_K) || defined(FC_F32_ABLATE_POLY) ||  \
     defined(FC_ABLATE_PUSH) || defined(FC_ABLATE_STAGE) || defined(FC_ABLATE_OVERFLOW) || defined(FC_ABLATE_REDO) ||         \
     defined(FC_RB_NOSTAGE) || defined(FC_RB_NOPASS1) || defined(FC_RB_NOPASS2) || defined(FC_RB_NOJACOBI) ||                \
     defined(FC_RB_NOCOMPUTE) || defined(FC_RB_ROWS_FROM_LDS) || defined(FC_RB_SAMEROW) || defined(FC_RB_WPS) ||              \
     defined(FC_RB_AHEAD) || defined(FC_RB_CHUNK) || defined(FC_H2_WGS) || defined(FC_F32_WGS) || defined(FC_REFINE_ROUNDS) || defined(FC_V2_ALIGN) || defined(FC_STAGE_PAIRS_F32) || defined(FC_TS_WGS) || defined(FC_TS_TURN))
#error "a tuning switch (FC_*TIMELINE / FC_*ABLATE* / FC_RB_* / FC_TFD_STAMPS ...) without -DFC_TUNING_BUILD: see fc_tuning.h"
#endif
