// cupoch/cupoch.h -- aggregate header of the ICP path (reference: src/cupoch/cupoch.h)
#pragma once
#include "cupoch/camera/pinhole_camera_intrinsic.h"
#include "cupoch/geometry/image.h"
#include "cupoch/geometry/pointcloud.h"
#include "cupoch/kinfu/kinfu.h"
#include "cupoch/odometry/odometry.h"
#include "cupoch/knn/kdtree_flann.h"
#include "cupoch/knn/kdtree_search_param.h"
#include "cupoch/registration/generalized_icp.h"
#include "cupoch/registration/registration.h"
#include "cupoch/registration/transformation_estimation.h"
#include "cupoch/utility/device_vector.h"
#include "cupoch/utility/eigen.h"
