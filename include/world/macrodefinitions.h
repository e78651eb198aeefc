/* macrodefinitions.h -- C-linkage brackets for the WORLD-compatible headers.
 * Mirrors the role of externs/WORLD_v2/src/world/macrodefinitions.h:66-74. */
#ifndef WORLD_MI355_MACRODEFINITIONS_H_
#define WORLD_MI355_MACRODEFINITIONS_H_
#ifdef __cplusplus
#define WORLD_BEGIN_C_DECLS extern "C" {
#define WORLD_END_C_DECLS }
#else
#define WORLD_BEGIN_C_DECLS
#define WORLD_END_C_DECLS
#endif
#endif
