// o3dr_host.h — C++ host mirror of the reference's `Pose` class for the reconstruction hot path.
//
// Same member-function names, argument meaning and error behaviour as pose.h:198,199,216,231 of the
// reference; the bodies call libo3dr (include/o3dr.h) instead of OpenCV/PCL.  Everything outside the
// hot path that the CLI needs to run end to end (flag parsing, calibration/pose/time CSV readers,
// timestamp binding, generateTmat, the variance gate, PNG/PLY I/O) is restated here in plain C++:
// it is control plane, one call per frame or per run, and stays on the host.  Pose estimation (ORB,
// matching, ICP), visualisation and the mesh/segment tools are not part of this build; the CLI runs
// with the recorded MAVLink poses (the reference's --only_MAVLink mode, pose_functions.cpp:232-236).
#pragma once
#include <array>
#include <cstdint>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/o3dr.h"

namespace o3dr_host {

typedef o3dr_point PointXYZRGB;                 // 16 B: x,y,z + a<<24|r<<16|g<<8|b
typedef std::array<float, 16> Matrix4;          // row-major 4x4 (the reference's Eigen Matrix4f, by value)

struct PointCloud {                             // the slice of pcl::PointCloud the hot path uses
    typedef std::shared_ptr<PointCloud> Ptr;
    std::vector<PointXYZRGB> points;
    bool is_dense = true;
    size_t size() const { return points.size(); }
};

struct Image8 {                                 // cv::Mat stand-in: 8-bit, 1 or 3 interleaved channels (B,G,R)
    int rows = 0, cols = 0, channels = 0;
    std::vector<uint8_t> data;
    bool empty() const { return data.empty(); }
    int64_t pitch() const { return (int64_t)cols * channels; }
};

// ---- I/O helpers (png_io.cpp, ply_io.cpp) ------------------------------------------------------------
// cv::imread(path) / cv::imread(path, IMREAD_GRAYSCALE) for 8-bit non-interlaced PNGs; empty on failure
Image8 read_png(const std::string& path, bool grayscale);
// pcl::io::savePLYFileBinary layout (x,y,z float + r,g,b uchar, then one `camera` element)
bool save_ply_binary(const std::string& path, const PointCloud& cloud);
// pcl::PLYReader for the files save_ply_binary writes (and build/cloud.ply)
bool read_ply(const std::string& path, PointCloud& cloud);

class RawImageData {  // pose.h:54-70
public:
    int img_num = 0;
    Image8 rgb_image, disparity_image;
    double time = 0, tx = 0, ty = 0, tz = 0, qx = 0, qy = 0, qz = 0, qw = 1;
};

class ImageData {  // pose.h:73-85 (the hot-path fields)
public:
    RawImageData* raw_img_data_ptr = nullptr;
    std::vector<float> keypoints_xy;  // features.keypoints pt.x,pt.y pairs; empty: no ORB in this build
    Matrix4 t_mat_MAVLink{}, t_mat_FeatureMatched{};
};

class Pose {
public:
    Pose(int argc, char* argv[]);  // like the reference, the whole program runs inside the constructor
    ~Pose();

    // ---- the `Pose` members the hot path reads, reference defaults (pose.h:92-175) --------------------
    double minDisparity = 64;
    int boundingBox = 20;
    int rows = 0, cols = 0, cols_start_aft_cutout = 0;
    int jump_pixels = 10;
    int seq_len = -1;
    int blur_kernel = 1;
    unsigned int min_points_per_voxel = 1;
    double voxel_size = 0.1;
    int cutout_ratio = 8;
    bool dont_downsample = false, downsample = false, log_stuff = false, only_MAVLink = true, dont_icp = true;
    bool reference_fanout = false;  // run A6 through createAndTransformPtCloud on 7 host threads
    bool sor = true;                // statistical outlier removal of the per-frame path (pose_functions.cpp:1673-1686:
                                    // always on in the reference when jump_pixels > 0); `--sor 0` switches it off
    std::string keypointsPrefix;    // --keypoints_dir: <img_num>.txt with one "x y" pair per line (KeyPoint::pt of the
                                    // frame's ORB features, which this build does not compute); empty = no keypoints
    std::array<double, 16> Q{};
    std::string calib_file = "cam13calib.yml";
    std::string dataFilesPrefix = "data_files/", imagePrefix = "images/", disparityPrefix = "disparities/";
    std::string outputPrefix = "output/";
    std::string read_PLY_filename0;
    int device_id = 0;
    int n_gpus = 1;                 // --gpus N: frames sharded over devices device_id .. device_id+N-1, one host thread and
                                    // one context each, merged through o3dr_merge_partitioned (RCCL)
    bool partitioned_merge = false; // --partitioned_merge: take that path with one GPU as well

    std::vector<RawImageData> rawImageDataVec;
    std::vector<ImageData> acceptedImageDataVec;

    // ---- the four entry points (pose.h:198,199,216,231) ------------------------------------------------
    void createSingleImgPtCloud(int accepted_img_index, PointCloud::Ptr cloudrgb);
    void transformPtCloud(PointCloud::Ptr cloudrgb, PointCloud::Ptr transformed_cloudrgb, Matrix4 transform);
    PointCloud::Ptr downsamplePtCloud(PointCloud::Ptr& cloudrgb, bool combinedPtCloud);
    void createAndTransformPtCloud(int accepted_img_index, PointCloud::Ptr& cloudrgb_return);

    // ---- control plane around them ---------------------------------------------------------------------
    Matrix4 generateTmat(int current_idx);                 // pose_functions.cpp:1178-1356
    double getMean(const Image8& disp_img);                // pose_functions.cpp:987-1005
    double getVariance(const Image8& disp_img);            // pose_functions.cpp:1007-1028
    int parseCmdArgs(int argc, char** argv);               // pose_functions.cpp:70-307 (hot-path flags)
    void printUsage();
    void readCalibFile();                                  // pose_functions.cpp:467-476
    void readPoseFile();                                   // pose_functions.cpp:478-506
    void populateData();                                   // pose_functions.cpp:624-744
    int binarySearchImageTime(int l, int r, int imageNumber);
    int binarySearchUsingTime(const std::vector<double>& seq, int l, int r, double time);
    void save_pt_cloud_to_PLY_File(PointCloud::Ptr cloudrgb, std::string& writePath);
    PointCloud::Ptr read_PLY_File(std::string point_cloud_filename);

private:
    o3dr_ctx* ctx_for_this_thread();
    void push_params(o3dr_ctx* c);
    void run_reconstruction();
    void run_sharded(PointCloud::Ptr cloud_small);  // --gpus N
    int first_img_num = -1, last_img_num = -1;
    bool run3d_reconstruction = true;
    std::vector<std::vector<double>> pose_data, images_times_data;
    std::vector<double> pose_times_seq, images_times_seq;
    std::ofstream log_file;
    std::vector<o3dr_ctx*> all_ctx;
};

}  // namespace o3dr_host
