#pragma once
#include "gg_device.h"
