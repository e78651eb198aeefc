/* world_oracle_vibrato.c -- placeholder translation unit; filled by the vibrato restatement. */
#include "world_oracle.h"
