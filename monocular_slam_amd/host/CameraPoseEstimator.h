// forwards to Stages.h (all three stage classes are declared there)
#pragma once
#include "Stages.h"
