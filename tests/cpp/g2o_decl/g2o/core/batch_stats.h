#include "../../g2o_decl_all.h"   /* test-only declarations, see g2o_decl_all.h */
