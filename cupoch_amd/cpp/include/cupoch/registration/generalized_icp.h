// cupoch/registration/generalized_icp.h -- declarations live in registration.h /
// transformation_estimation.h; kept so that reference #includes resolve.
#pragma once
#include "cupoch/registration/registration.h"
