#include "sync_oracle.h"
