/* macrodefinitions.h -- C-linkage brackets for the WORLD-compatible headers.
 * Mirrors the role of externs/WORLD_v2/src/world/macrodefinitions.h:66-74.
 * libworld_mi355.so is built with -fvisibility=hidden: what these brackets enclose is its exported ABI (the
 * pragma marks the declarations, and with them the definitions, as visible; harmless for a caller). */
#ifndef WORLD_MI355_MACRODEFINITIONS_H_
#define WORLD_MI355_MACRODEFINITIONS_H_
#if defined(__GNUC__) || defined(__clang__)
#define WORLD_MI355_EXPORT_BEGIN _Pragma("GCC visibility push(default)")
#define WORLD_MI355_EXPORT_END _Pragma("GCC visibility pop")
#else
#define WORLD_MI355_EXPORT_BEGIN
#define WORLD_MI355_EXPORT_END
#endif
#ifdef __cplusplus
#define WORLD_BEGIN_C_DECLS extern "C" { WORLD_MI355_EXPORT_BEGIN
#define WORLD_END_C_DECLS WORLD_MI355_EXPORT_END }
#else
#define WORLD_BEGIN_C_DECLS WORLD_MI355_EXPORT_BEGIN
#define WORLD_END_C_DECLS WORLD_MI355_EXPORT_END
#endif
#endif
