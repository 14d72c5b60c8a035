// test host: include forwarder (see ../../g2o_mini.h)
#include "../../g2o_mini.h"
