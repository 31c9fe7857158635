// pose.cpp — host side of the hot path: the reference's `Pose` entry points on top of libo3dr.
// Reference lines restated are cited at each function (paths relative to the reference tree).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "o3dr_host.h"

using namespace std;

namespace o3dr_host {

static void chk(int rc, const char* what)
{
    if (rc != O3DR_OK) throw runtime_error(string(what) + ": " + o3dr_last_error());
}

// ------------------------------------------------------------------------------------------------
// contexts: the reference calls createAndTransformPtCloud from up to 7 threads (pose.cpp:392-413);
// a libo3dr context is single-threaded, so each calling thread gets its own.
// ------------------------------------------------------------------------------------------------
static mutex g_ctx_mu;
static thread_local o3dr_ctx* tl_ctx = nullptr;

o3dr_ctx* Pose::ctx_for_this_thread()
{
    if (!tl_ctx) {
        chk(o3dr_ctx_create(device_id, &tl_ctx), "o3dr_ctx_create");
        lock_guard<mutex> lk(g_ctx_mu);
        all_ctx.push_back(tl_ctx);
    }
    push_params(tl_ctx);
    return tl_ctx;
}

void Pose::push_params(o3dr_ctx* c)
{
    o3dr_params p;
    o3dr_default_params(&p);
    p.min_disparity = minDisparity;
    p.voxel_size = voxel_size;
    p.bounding_box = boundingBox;
    p.cutout_ratio = cutout_ratio;
    p.jump_pixels = jump_pixels;
    p.min_points_per_voxel = min_points_per_voxel;
    p.dont_downsample = dont_downsample ? 1 : 0;
    p.sor_enable = sor ? 1 : 0;
    p.blur_kernel = blur_kernel;  // > 1: bilateral filter on the disparity image first (pose_functions.cpp:1040-1047)
    chk(o3dr_set_params(c, &p), "o3dr_set_params");
    chk(o3dr_set_camera(c, Q.data()), "o3dr_set_camera");
}

Pose::~Pose()
{
    for (o3dr_ctx* c : all_ctx) o3dr_ctx_destroy(c);
    if (tl_ctx) tl_ctx = nullptr;
}

// ------------------------------------------------------------------------------------------------
// the four entry points
// ------------------------------------------------------------------------------------------------
// pose.h:198 / pose_functions.cpp:1030-1134
void Pose::createSingleImgPtCloud(int accepted_img_index, PointCloud::Ptr cloudrgb)
{
    cloudrgb->is_dense = true;
    const ImageData& im = acceptedImageDataVec[accepted_img_index];
    const RawImageData& raw = *im.raw_img_data_ptr;
    o3dr_ctx* c = ctx_for_this_thread();
    const int n_kp = (int)(im.keypoints_xy.size() / 2);
    const int64_t cap = o3dr_max_points(c, rows, cols) + n_kp;
    cloudrgb->points.resize((size_t)(cap > 0 ? cap : 1));
    int64_t n = 0;
    chk(o3dr_create_single_img_pt_cloud(c, raw.disparity_image.data.data(), raw.disparity_image.pitch(),
                                        raw.rgb_image.data.data(), raw.rgb_image.pitch(), rows, cols,
                                        n_kp ? im.keypoints_xy.data() : nullptr, n_kp, cloudrgb->points.data(), cap, &n,
                                        O3DR_MEM_HOST),
        "createSingleImgPtCloud");
    cloudrgb->points.resize((size_t)n);
    cout << " " << raw.img_num << std::flush;  // :1131-1133
    if (log_file.is_open()) log_file << " " << raw.img_num << "/" << cloudrgb->points.size() << std::flush;
}

// pose.h:199 / pose_functions.cpp:1358-1362
void Pose::transformPtCloud(PointCloud::Ptr cloudrgb, PointCloud::Ptr transformed_cloudrgb, Matrix4 transform)
{
    transformed_cloudrgb->points.resize(cloudrgb->points.size());
    transformed_cloudrgb->is_dense = cloudrgb->is_dense;
    chk(o3dr_transform_pt_cloud(ctx_for_this_thread(), cloudrgb->points.data(), (int64_t)cloudrgb->points.size(),
                                transform.data(), transformed_cloudrgb->points.data(), O3DR_MEM_HOST),
        "transformPtCloud");
}

// pose.h:216 / pose_functions.cpp:1654-1709
PointCloud::Ptr Pose::downsamplePtCloud(PointCloud::Ptr& cloudrgb, bool combinedPtCloud)
{
    PointCloud::Ptr out(new PointCloud());
    const int64_t n_in = (int64_t)cloudrgb->points.size();
    out->points.resize((size_t)(n_in > 0 ? n_in : 1));
    int64_t n = 0;
    uint32_t status = 0;
    chk(o3dr_downsample_pt_cloud(ctx_for_this_thread(), cloudrgb->points.data(), n_in, combinedPtCloud ? 1 : 0,
                                 out->points.data(), n_in > 0 ? n_in : 1, &n, &status, O3DR_MEM_HOST),
        "downsamplePtCloud");
    if (status & O3DR_STATUS_VOXEL_OVERFLOW)  // PCL_WARN of VoxelGrid::applyFilter
        cerr << "[pcl::VoxelGrid::applyFilter] Leaf size is too small for the input dataset. Integer indices would overflow."
             << endl;
    out->points.resize((size_t)n);
    return out;
}

// pose.h:231 / pose.cpp:596-636: every exception is caught and printed, the output cloud is left empty
void Pose::createAndTransformPtCloud(int accepted_img_index, PointCloud::Ptr& cloudrgb_return)
{
    try {
        const ImageData& im = acceptedImageDataVec[accepted_img_index];
        const RawImageData& raw = *im.raw_img_data_ptr;
        o3dr_ctx* c = ctx_for_this_thread();
        const int n_kp = (int)(im.keypoints_xy.size() / 2);
        const int64_t cap = o3dr_max_points(c, rows, cols) + n_kp;
        cloudrgb_return->points.resize((size_t)(cap > 0 ? cap : 1));
        int64_t n = 0;
        uint32_t status = 0;
        const int rc = o3dr_create_and_transform_pt_cloud(
            c, raw.disparity_image.data.data(), raw.disparity_image.pitch(), raw.rgb_image.data.data(),
            raw.rgb_image.pitch(), rows, cols, im.t_mat_FeatureMatched.data(), n_kp ? im.keypoints_xy.data() : nullptr, n_kp,
            cloudrgb_return->points.data(), cap, &n, &status, O3DR_MEM_HOST);
        cloudrgb_return->points.resize((size_t)(rc == O3DR_OK ? n : 0));
        if (rc != O3DR_OK) throw runtime_error(o3dr_last_error());
        cout << " " << raw.img_num << std::flush;
        if (log_file.is_open()) log_file << " " << raw.img_num << "/" << n << std::flush;
    } catch (const std::exception& e) {
        cout << "Exception caught in thread with accepted_img_index=" << accepted_img_index << endl;
        cout << e.what() << endl;
        cloudrgb_return->points.clear();
    } catch (...) {
        cout << "General Exception caught in thread with accepted_img_index=" << accepted_img_index << endl;
        cloudrgb_return->points.clear();
    }
}

// ------------------------------------------------------------------------------------------------
// control plane: pose matrix, gates, readers
// ------------------------------------------------------------------------------------------------
static Matrix4 mul4(const Matrix4& a, const Matrix4& b)
{  // float 4x4 product, each coefficient ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j
    Matrix4 o{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a[i * 4] * b[j];
            for (int k = 1; k < 4; ++k) s = s + a[i * 4 + k] * b[k * 4 + j];
            o[i * 4 + j] = s;
        }
    return o;
}
static Matrix4 ident()
{
    Matrix4 m{};
    m[0] = m[5] = m[10] = m[15] = 1.f;
    return m;
}

// pose_functions.cpp:1178-1356: camera mounting (pose.h:142-147) x quaternion x translation, as float
// 4x4 factors multiplied left to right (:1341).  Eigen may order/fuse the inner sums differently from
// this plain loop (agreement ~1e-6).
Matrix4 Pose::generateTmat(int current_idx)
{
    const double PI = 3.141592653589793238463;
    const double theta_xi = -1.1408 * PI / 180, theta_yi = 1.1945 * PI / 180;
    const double trans_x_hi = -0.300, trans_y_hi = -0.040, trans_z_hi = -0.350;
    Matrix4 r_xi = ident();
    r_xi[5] = (float)cos(theta_xi); r_xi[6] = (float)-sin(theta_xi); r_xi[9] = (float)sin(theta_xi); r_xi[10] = (float)cos(theta_xi);
    Matrix4 r_yi = ident();
    r_yi[0] = (float)cos(theta_yi); r_yi[2] = (float)sin(theta_yi); r_yi[8] = (float)-sin(theta_yi); r_yi[10] = (float)cos(theta_yi);
    Matrix4 r_invert_i = ident();
    r_invert_i[5] = -1.f; r_invert_i[10] = -1.f;
    Matrix4 r_invert_y = ident();
    r_invert_y[5] = -1.f;
    Matrix4 t_hi = ident();
    t_hi[3] = (float)trans_x_hi; t_hi[7] = (float)trans_y_hi; t_hi[11] = (float)trans_z_hi;
    Matrix4 r_flip_xy{};
    r_flip_xy[4] = 1.f; r_flip_xy[1] = 1.f; r_flip_xy[10] = 1.f; r_flip_xy[15] = 1.f;

    const RawImageData& r = rawImageDataVec[current_idx];
    const double qx = r.qx, qy = r.qy, qz = r.qz, qw = r.qw;
    const double sqw = qw * qw, sqx = qx * qx, sqy = qy * qy, sqz = qz * qz;
    if (sqw + sqx + sqy + sqz < 0.99 || sqw + sqx + sqy + sqz > 1.01)
        throw "Exception: Sum of squares of quaternion values should be 1! i.e., quaternion should be homogeneous!";
    double rot[3][3];
    rot[0][0] = sqx - sqy - sqz + sqw;
    rot[1][1] = -sqx + sqy - sqz + sqw;
    rot[2][2] = -sqx - sqy + sqz + sqw;
    double t1 = qx * qy, t2 = qz * qw;
    rot[0][1] = 2.0 * (t1 + t2);
    rot[1][0] = 2.0 * (t1 - t2);
    t1 = qx * qz; t2 = qy * qw;
    rot[0][2] = 2.0 * (t1 - t2);
    rot[2][0] = 2.0 * (t1 + t2);
    t1 = qy * qz; t2 = qx * qw;
    rot[1][2] = 2.0 * (t1 + t2);
    rot[2][1] = 2.0 * (t1 - t2);
    Matrix4 r_wh = ident();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r_wh[i * 4 + j] = (float)rot[j][i];  // rot = rot.t()
    Matrix4 t_wh = ident();
    t_wh[3] = (float)r.tx; t_wh[7] = (float)r.ty; t_wh[11] = (float)r.tz;
    Matrix4 m = t_wh;
    const Matrix4* chain[] = {&r_wh, &r_invert_y, &r_flip_xy, &t_hi, &r_invert_i, &r_yi, &r_xi};
    for (const Matrix4* f : chain) m = mul4(m, *f);
    return m;
}

// pose_functions.cpp:987-1005
double Pose::getMean(const Image8& disp_img)
{
    double sum = 0.0;
    for (int y = boundingBox; y < rows - boundingBox; ++y)
        for (int x = cols_start_aft_cutout; x < cols - boundingBox; ++x) {
            const double d = (double)disp_img.data[(size_t)y * disp_img.cols + x];
            if (d > minDisparity) sum += d;
        }
    return sum / ((rows - 2 * boundingBox) * (cols - boundingBox - cols_start_aft_cutout));
}
// pose_functions.cpp:1007-1028
double Pose::getVariance(const Image8& disp_img)
{
    const double mean = getMean(disp_img);
    double temp = 0;
    for (int y = boundingBox; y < rows - boundingBox; ++y)
        for (int x = cols_start_aft_cutout; x < cols - boundingBox; ++x) {
            const double d = (double)disp_img.data[(size_t)y * disp_img.cols + x];
            if (d > minDisparity) temp += (d - mean) * (d - mean);
        }
    return temp / ((rows - 2 * boundingBox) * (cols - boundingBox - cols_start_aft_cutout) - 1);
}

// CSV of doubles, one record per line (pose_functions.cpp:351-397)
static vector<vector<double>> read_csv(const string& path)
{
    ifstream f(path);
    if (!f.is_open()) throw runtime_error("Exception: Could not open " + path);
    vector<vector<double>> data;
    string line;
    while (getline(f, line)) {
        vector<double> rec;
        stringstream ss(line);
        string field;
        while (getline(ss, field, ',')) rec.push_back(strtod(field.c_str(), nullptr));
        data.push_back(rec);
    }
    return data;
}

// pose_functions.cpp:402-426
int Pose::binarySearchImageTime(int l, int r, int imageNumber)
{
    while (r >= l) {
        const int mid = l + (r - l) / 2;
        const int v = (int)images_times_data[mid][0];
        if (v == imageNumber) return mid;
        if (v > imageNumber) r = mid - 1; else l = mid + 1;
    }
    throw "Exception: binarySearchImageTime: unsuccessful search!";
}
// pose_functions.cpp:431-465: returns an index whose NEIGHBOURS bracket `time`
int Pose::binarySearchUsingTime(const vector<double>& seq, int l, int r, double time)
{
    while (r >= l) {
        const int mid = l + (r - l) / 2;
        if (mid > 0 && mid < (int)seq.size() - 1) {
            if (seq[mid - 1] < time && seq[mid + 1] > time) return mid;
        } else if (mid == 0) {
            return 0;
        } else {
            return (int)seq.size() - 1;
        }
        if (seq[mid] > time) r = mid - 1; else l = mid + 1;
    }
    throw "Exception: binarySearchUsingTime: unsuccessful search!";
}

// pose_functions.cpp:467-476 — `Q: !!opencv-matrix ... data: [ 16 doubles ]` of an OpenCV YAML file
void Pose::readCalibFile()
{
    ifstream f(dataFilesPrefix + calib_file);
    if (!f.is_open()) throw "Exception: could not read Q matrix";
    stringstream ss;
    ss << f.rdbuf();
    const string txt = ss.str();
    size_t p = txt.find("\nQ:");
    if (p == string::npos) p = txt.rfind("Q:");
    if (p == string::npos) throw "Exception: could not read Q matrix";
    p = txt.find("data:", p);
    const size_t a = txt.find('[', p), b = txt.find(']', a);
    if (p == string::npos || a == string::npos || b == string::npos) throw "Exception: could not read Q matrix";
    string body = txt.substr(a + 1, b - a - 1);
    for (char& ch : body)
        if (ch == ',' || ch == '\n') ch = ' ';
    stringstream vs(body);
    for (int i = 0; i < 16; ++i)
        if (!(vs >> Q[i])) throw "Exception: could not read Q matrix";
    cout << "read calib file." << endl;
}

// pose_functions.cpp:478-506
void Pose::readPoseFile()
{
    pose_data = read_csv(dataFilesPrefix + "pose.txt");
    pose_times_seq.clear();
    for (auto& r : pose_data) pose_times_seq.push_back(r.size() > 2 ? r[2] : 0.0);
    images_times_data = read_csv(dataFilesPrefix + "images.txt");
    images_times_seq.clear();
    for (auto& r : images_times_data) images_times_seq.push_back(r.size() > 2 ? r[2] : 0.0);
    cout << "Your images_times file contains " << images_times_data.size() << " records.\n";
    cout << "Your pose_data file contains " << pose_data.size() << " records.\n";
}

// pose_functions.cpp:624-744: the reference spreads imread over 7+7 threads; so does this
void Pose::populateData()
{
    readCalibFile();
    if (log_stuff) log_file.open(outputPrefix + "log.txt", ios::out);
    const int n = (int)rawImageDataVec.size();
    const int n_threads = 7;
    vector<thread> pool;
    mutex mu;
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t]() {
            for (int i = t; i < n; i += n_threads) {
                RawImageData& r = rawImageDataVec[i];
                r.rgb_image = read_png(imagePrefix + to_string(r.img_num) + ".png", false);       // :523-536
                r.disparity_image = read_png(disparityPrefix + to_string(r.img_num) + ".png", true);  // :546-585
                try {
                    const int it = binarySearchImageTime(0, (int)images_times_seq.size() - 1, r.img_num);
                    const int ip = binarySearchUsingTime(pose_times_seq, 0, (int)pose_times_seq.size() - 1, images_times_seq[it]);
                    r.time = images_times_seq[it];
                    const vector<double>& p = pose_data[ip];  // pose.h:140 tx_ind=3 .. qw_ind=9
                    r.tx = p[3]; r.ty = p[4]; r.tz = p[5]; r.qx = p[6]; r.qy = p[7]; r.qz = p[8]; r.qw = p[9];
                } catch (...) {
                    lock_guard<mutex> lk(mu);
                    cout << " no_pose_for_" << r.img_num << " " << flush;
                    r.disparity_image = Image8();
                }
                lock_guard<mutex> lk(mu);
                cout << (r.rgb_image.empty() ? " cannot_read_i" : " i") << r.img_num << (r.disparity_image.empty() ? " cannot_read_d" : " d")
                     << r.img_num << " " << flush;
            }
        });
    for (thread& t : pool) t.join();
    cout << endl;
    for (const RawImageData& r : rawImageDataVec)
        if (!r.disparity_image.empty()) {  // rows/cols from the first readable image (:635-638)
            rows = r.disparity_image.rows;
            cols = r.disparity_image.cols;
            cols_start_aft_cutout = (int)(cols / cutout_ratio);
            break;
        }
}

void Pose::save_pt_cloud_to_PLY_File(PointCloud::Ptr cloudrgb, string& writePath)
{
    if (!save_ply_binary(writePath, *cloudrgb)) throw runtime_error("could not write " + writePath);
    cerr << "Saved Point Cloud with " << cloudrgb->points.size() << " data points to " << writePath << endl;
}
PointCloud::Ptr Pose::read_PLY_File(string point_cloud_filename)
{
    cout << "Reading PLY file..." << endl;
    PointCloud::Ptr c(new PointCloud());
    if (!read_ply(point_cloud_filename, *c)) throw runtime_error("could not read " + point_cloud_filename);
    cout << "Read PLY file!" << endl;
    return c;
}

// ------------------------------------------------------------------------------------------------
// CLI (pose_functions.cpp:3-307): the hot-path flags keep their names and meaning; the three data
// directories, hard-coded absolute paths in the reference (pose.h:134-137), are flags here.
// ------------------------------------------------------------------------------------------------
void Pose::printUsage()
{
    cout << "./pose first_img last_img [--voxel_size m] [--jump_pixels n] [--min_points_per_voxel n] [--seq_len n]\n"
            "       [--dont_downsample] [--log 0|1] [--only_MAVLink] [--dont_icp] [--reference_fanout] [--sor 0|1]\n"
            "       [--data_dir d/] [--image_dir d/] [--disparity_dir d/] [--output_dir d/] [--calib_file f] [--device n]\n"
            "       [--keypoints_dir d/]   (d/<img_num>.txt: one \"x y\" keypoint per line; used iff jump_pixels != 1)\n"
            "       [--gpus N]   (frames sharded over N GPUs from --device on, one host thread each; the final merge is exchanged\n"
            "                     over RCCL and equals the one-GPU result bit for bit)  [--partitioned_merge]  (same path, N = 1)\n"
            "--sor defaults to 1: like the reference, every per-frame cloud goes through StatisticalOutlierRemoval(50, 1.0)\n"
            "before its voxel grid when jump_pixels > 0.\n"
            "./pose --downsample file.ply [--voxel_size m] [--min_points_per_voxel n]\n"
            "Pose estimation (ORB matching, ICP), visualisation and the mesh/segment tools are not part of this build.\n";
}

int Pose::parseCmdArgs(int argc, char** argv)
{
    if (argc == 1) {
        printUsage();
        return -1;
    }
    int n_imgs = 0;
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) throw runtime_error(string("missing value after ") + argv[i]);
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        const string a = argv[i];
        if (a == "--help" || a == "/?") { printUsage(); return -1; }
        else if (a == "--downsample") { downsample = true; run3d_reconstruction = false; read_PLY_filename0 = need(i); }
        else if (a == "--voxel_size") voxel_size = atof(need(i));
        else if (a == "--min_points_per_voxel") min_points_per_voxel = (unsigned)atoi(need(i));
        else if (a == "--jump_pixels") jump_pixels = atoi(need(i));
        else if (a == "--seq_len") seq_len = atoi(need(i));
        else if (a == "--blur_kernel") blur_kernel = atoi(need(i));
        else if (a == "--log") log_stuff = atoi(need(i)) != 0;
        else if (a == "--dont_downsample") dont_downsample = true;
        else if (a == "--only_MAVLink") only_MAVLink = true;
        else if (a == "--dont_icp") dont_icp = true;
        else if (a == "--reference_fanout") reference_fanout = true;
        else if (a == "--sor") sor = atoi(need(i)) != 0;
        else if (a == "--data_dir") dataFilesPrefix = need(i);
        else if (a == "--image_dir") imagePrefix = need(i);
        else if (a == "--disparity_dir") disparityPrefix = need(i);
        else if (a == "--output_dir") outputPrefix = need(i);
        else if (a == "--keypoints_dir") keypointsPrefix = need(i);
        else if (a == "--calib_file") calib_file = need(i);
        else if (a == "--device") device_id = atoi(need(i));
        else if (a == "--gpus") n_gpus = atoi(need(i));
        else if (a == "--partitioned_merge") partitioned_merge = true;
        else if (a == "--dist_nearby" || a == "--search_radius" || a == "--range_width") { need(i); }
        else if (a == "--preview" || a == "--use_segment_labels" || a == "--segment_cloud" || a == "--displayUAVPositions" ||
                 a == "--test_bad_data_rejection")
            cout << a << ": outside the hot path, ignored in this build" << endl;
        else if (a.rfind("--", 0) == 0) throw runtime_error("unknown flag " + a);
        else {  // positional image numbers (:247-256)
            if (first_img_num == -1) first_img_num = atoi(argv[i]); else last_img_num = atoi(argv[i]);
            ++n_imgs;
        }
    }
    if (run3d_reconstruction) {
        if (n_imgs == 0) throw runtime_error("first and last image number are required");
        if (last_img_num < first_img_num) last_img_num = first_img_num;
        readPoseFile();
        n_imgs = last_img_num - first_img_num + 1;  // :300-303
        rawImageDataVec = vector<RawImageData>((size_t)n_imgs);
        for (int i = 0; i < n_imgs; ++i) rawImageDataVec[i].img_num = first_img_num + i;
    }
    return 0;
}

// pose.cpp:23-565 restricted to the hot path
Pose::Pose(int argc, char* argv[])
{
    if (parseCmdArgs(argc, argv) != 0) return;
    if (downsample) {  // pose.cpp:71-87

        Q = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // the tool needs no camera
        PointCloud::Ptr cloud = read_PLY_File(read_PLY_filename0);
        PointCloud::Ptr small = downsamplePtCloud(cloud, true);
        string out = read_PLY_filename0;
        const size_t slash = out.find_last_of('/');
        out = (slash == string::npos ? string() : out.substr(0, slash + 1)) + "downsampled_" +
              (slash == string::npos ? out : out.substr(slash + 1));
        save_pt_cloud_to_PLY_File(small, out);
        return;
    }
    if (!run3d_reconstruction) return;
    populateData();
    if (rows == 0 || cols == 0 || cols_start_aft_cutout == 0)
        throw "Exception: some important values not set! rows/cols/cols_start_aft_cutout";
    run_reconstruction();
}

void Pose::run_reconstruction()
{
    typedef chrono::steady_clock clk;
    const auto app_start = clk::now();
    o3dr_ctx* c = ctx_for_this_thread();
    chk(o3dr_cloud_big_reset(c), "cloud_big_reset");
    PointCloud::Ptr cloud_big_host(new PointCloud());  // only used by --reference_fanout
    const int last_idx = (int)rawImageDataVec.size() - 1;
    const int cycle_len = seq_len > 0 ? seq_len : (int)rawImageDataVec.size();
    int current_idx = 0, cycle = 0;
    acceptedImageDataVec.reserve(rawImageDataVec.size());
    if (!sor && jump_pixels > 0)
        cout << "NOTE: --sor 0: the per-frame StatisticalOutlierRemoval of the reference (pose_functions.cpp:1673-1686) is OFF;\n"
                "      the clouds differ from the reference's for the same command line." << endl;
    cout << "\n\nProgram Start!" << endl;
    while (current_idx <= last_idx) {
        cout << "\nCycle " << cycle << endl;
        const size_t first_accepted = acceptedImageDataVec.size();
        int images_in_cycle = 0;
        while (images_in_cycle < cycle_len && current_idx <= last_idx) {  // pose.cpp:162-255
            RawImageData& r = rawImageDataVec[current_idx];
            if (r.rgb_image.empty()) { cout << r.img_num << " could not read rgb image. \tRejected!" << endl; current_idx++; continue; }
            if (r.disparity_image.empty()) { cout << r.img_num << " could not read disparity image. \tRejected!" << endl; current_idx++; continue; }
            const double var = getVariance(r.disparity_image);
            cout << r.img_num << " " << flush;
            if (var > 5) { cout << " disp_img_var = " << var << " > 5.\tRejected!" << endl; current_idx++; continue; }
            ImageData d;
            d.raw_img_data_ptr = &r;
            d.t_mat_MAVLink = generateTmat(current_idx);
            d.t_mat_FeatureMatched = d.t_mat_MAVLink;  // --only_MAVLink, pose.cpp:238
            if (!keypointsPrefix.empty()) {
                // the list the reference's ORB stage would leave in features.keypoints (pose_functions.cpp:1057-1061)
                ifstream kf(keypointsPrefix + to_string(r.img_num) + ".txt");
                float kx, ky;
                while (kf >> kx >> ky) {
                    d.keypoints_xy.push_back(kx);
                    d.keypoints_xy.push_back(ky);
                }
            }
            acceptedImageDataVec.push_back(d);
            cout << "\tAccepted!" << endl;
            current_idx++;
            images_in_cycle++;
        }
        // ---- point cloud creation for this cycle (pose.cpp:365-434) -------------------------------
        const auto t3 = clk::now();
        cout << "Adding Point Cloud number/points ";
        const size_t n_acc = acceptedImageDataVec.size() - first_accepted;
        const bool sharded = n_gpus > 1 || partitioned_merge;
        if (sharded) {
            // frames are sharded over the GPUs as contiguous blocks of the WHOLE accepted list (global frame order is what
            // the merge sums in): the clouds are made after the last cycle
            cout << "(deferred: --gpus)";
        } else if (reference_fanout) {
            // the reference's own structure: batches of <= 7 threads, results appended in frame order
            for (size_t i0 = first_accepted; i0 < first_accepted + n_acc; i0 += 7) {
                const size_t nb = min<size_t>(7, first_accepted + n_acc - i0);
                vector<PointCloud::Ptr> clouds(nb);
                vector<thread> th;
                for (size_t k = 0; k < nb; ++k) {
                    clouds[k].reset(new PointCloud());
                    th.emplace_back([this, i0, k, &clouds]() { createAndTransformPtCloud((int)(i0 + k), clouds[k]); });
                }
                for (thread& t : th) t.join();
                for (size_t k = 0; k < nb; ++k)
                    cloud_big_host->points.insert(cloud_big_host->points.end(), clouds[k]->points.begin(), clouds[k]->points.end());
            }
        } else if (n_acc) {
            // MI355X-native form of the same loop: the whole cycle in one batched call, cloud_big in HBM
            const RawImageData& r0 = *acceptedImageDataVec[first_accepted].raw_img_data_ptr;
            const size_t dsz = r0.disparity_image.data.size(), csz = r0.rgb_image.data.size();
            vector<uint8_t> disp(dsz * n_acc), bgr(csz * n_acc);
            vector<float> poses(16 * n_acc), kp_xy;
            vector<int64_t> kp_off(n_acc + 1, 0);
            for (size_t k = 0; k < n_acc; ++k) {
                const ImageData& im = acceptedImageDataVec[first_accepted + k];
                memcpy(&disp[k * dsz], im.raw_img_data_ptr->disparity_image.data.data(), dsz);
                memcpy(&bgr[k * csz], im.raw_img_data_ptr->rgb_image.data.data(), csz);
                memcpy(&poses[16 * k], im.t_mat_FeatureMatched.data(), 64);
                kp_xy.insert(kp_xy.end(), im.keypoints_xy.begin(), im.keypoints_xy.end());
                kp_off[k + 1] = (int64_t)(kp_xy.size() / 2);
                cout << " " << im.raw_img_data_ptr->img_num << flush;
            }
            // page-locked frame stacks cross PCIe by DMA (best effort: pageable memory works too)
            const bool reg_disp = o3dr_host_register(disp.data(), (int64_t)disp.size()) == O3DR_OK;
            const bool reg_bgr = o3dr_host_register(bgr.data(), (int64_t)bgr.size()) == O3DR_OK;
            const int rc_acc = o3dr_accumulate_frames_kp(c, disp.data(), (int64_t)dsz, cols, bgr.data(), (int64_t)csz, 3 * (int64_t)cols, rows,
                                                         cols, poses.data(), (int32_t)n_acc, kp_xy.empty() ? nullptr : kp_xy.data(),
                                                         kp_xy.empty() ? nullptr : kp_off.data(), O3DR_MEM_HOST);
            const string why_acc = rc_acc != O3DR_OK ? o3dr_last_error() : "";
            if (reg_disp) (void)o3dr_host_unregister(disp.data());  // (only what was registered; also on the error path)
            if (reg_bgr) (void)o3dr_host_unregister(bgr.data());
            if (rc_acc != O3DR_OK) throw runtime_error("accumulate_frames: " + why_acc);
        }
        chk(o3dr_ctx_synchronize(c), "synchronize");
        const double dt = chrono::duration<double>(clk::now() - t3).count();
        cout << "\n\nPoint Cloud Creation time: " << dt << " sec" << endl;  // pose.cpp:429-431
        if (log_file.is_open()) log_file << "Point Cloud Creation time:\t\t\t" << dt << " sec" << endl;
        cycle++;
    }
    const double total = chrono::duration<double>(clk::now() - app_start).count();
    cout << "\nFinished Pose Estimation, total time: " << total << " sec at " << acceptedImageDataVec.size() / total << " fps" << endl;

    // ---- final merge + save (pose.cpp:527-540) ------------------------------------------------------
    PointCloud::Ptr cloud_small(new PointCloud());
    if (n_gpus > 1 || partitioned_merge) {
        run_sharded(cloud_small);
    } else if (reference_fanout) {
        cloud_small = dont_downsample ? cloud_big_host : downsamplePtCloud(cloud_big_host, true);
    } else {
        int64_t n_big = 0, n_small = 0;
        uint32_t st = 0;
        chk(o3dr_cloud_big_size(c, &n_big, &st), "cloud_big_size");
        cloud_small->points.resize((size_t)(n_big > 0 ? n_big : 1));
        if (!dont_downsample) cout << "downsample before saving..." << endl;
        chk(o3dr_finalize(c, cloud_small->points.data(), n_big > 0 ? n_big : 1, &n_small, &st, O3DR_MEM_HOST), "finalize");
        cloud_small->points.resize((size_t)n_small);
        if (st & O3DR_STATUS_VOXEL_OVERFLOW)
            cerr << "[pcl::VoxelGrid::applyFilter] Leaf size is too small for the input dataset. Integer indices would overflow." << endl;
        cout << "cloud_big " << n_big << " points -> cloud " << n_small << " points" << endl;
    }
    cout << "Saving point clouds..." << endl;
    string path = outputPrefix + "cloud.ply";
    save_pt_cloud_to_PLY_File(cloud_small, path);
}

// --gpus N: the fan-out of pose.cpp:392-413 over GPUs instead of threads of one CPU.  Rank g (host thread g, device
// device_id + g) takes the g-th contiguous block of the accepted frames through the batched call; o3dr_merge_partitioned
// then exchanges the per-frame voxels by index slice over RCCL, merges every slice on its GPU and gathers the result.
void Pose::run_sharded(PointCloud::Ptr cloud_small)
{
    typedef chrono::steady_clock clk;
    const auto t0 = clk::now();
    const size_t n_acc = acceptedImageDataVec.size();
    if (n_gpus < 1) throw runtime_error("--gpus must be at least 1");
    if (dont_downsample) throw runtime_error("--gpus / --partitioned_merge need the downsampling path (no --dont_downsample)");
    if (n_acc == 0) return;
    const RawImageData& r0 = *acceptedImageDataVec[0].raw_img_data_ptr;
    const size_t dsz = r0.disparity_image.data.size(), csz = r0.rgb_image.data.size();
    vector<uint8_t> disp(dsz * n_acc), bgr(csz * n_acc);
    vector<float> poses(16 * n_acc), kp_xy;
    vector<int64_t> kp_off(n_acc + 1, 0);
    for (size_t k = 0; k < n_acc; ++k) {
        const ImageData& im = acceptedImageDataVec[k];
        memcpy(&disp[k * dsz], im.raw_img_data_ptr->disparity_image.data.data(), dsz);
        memcpy(&bgr[k * csz], im.raw_img_data_ptr->rgb_image.data.data(), csz);
        memcpy(&poses[16 * k], im.t_mat_FeatureMatched.data(), 64);
        kp_xy.insert(kp_xy.end(), im.keypoints_xy.begin(), im.keypoints_xy.end());
        kp_off[k + 1] = (int64_t)(kp_xy.size() / 2);
    }
    // (best effort: pageable memory works too, through the runtime's bounce buffers; only what was registered is released)
    const bool reg_disp = o3dr_host_register(disp.data(), (int64_t)disp.size()) == O3DR_OK;
    const bool reg_bgr = o3dr_host_register(bgr.data(), (int64_t)bgr.size()) == O3DR_OK;
    const int W = n_gpus;
    vector<int32_t> devs((size_t)W);
    for (int g = 0; g < W; ++g) devs[(size_t)g] = device_id + g;
    vector<void*> comms((size_t)W, nullptr);
    chk(o3dr_comm_init_all(W, devs.data(), comms.data()), "o3dr_comm_init_all");
    // an upper bound of the merged cloud: no more cells than points, no more points than grid candidates + keypoints
    o3dr_ctx* c0 = ctx_for_this_thread();
    const int64_t cap = o3dr_max_points(c0, rows, cols) * (int64_t)n_acc + kp_off[n_acc] + 1;
    cloud_small->points.resize((size_t)cap);
    vector<string> errors((size_t)W);
    vector<int64_t> n_out((size_t)W, 0), n_total((size_t)W, 0);
    vector<uint32_t> st((size_t)W, 0);
    // Every rank's context exists before any rank starts: a rank without one could never enter the exchange, and its
    // peers would wait for it inside the first collective.  Past this point a failure on one rank is carried through the
    // collectives by o3dr_merge_partitioned itself (every rank returns together).
    vector<o3dr_ctx*> ctxs((size_t)W, nullptr);
    for (int g = 0; g < W; ++g) {
        if (o3dr_ctx_create(devs[(size_t)g], &ctxs[(size_t)g]) != O3DR_OK) {
            const string why = o3dr_last_error();
            for (int k = 0; k < g; ++k) (void)o3dr_ctx_destroy(ctxs[(size_t)k]);
            for (int k = 0; k < W; ++k) (void)o3dr_comm_destroy(comms[(size_t)k]);
            if (reg_disp) (void)o3dr_host_unregister(disp.data());
            if (reg_bgr) (void)o3dr_host_unregister(bgr.data());
            throw runtime_error("GPU " + to_string(devs[(size_t)g]) + ": o3dr_ctx_create: " + why);
        }
    }
    vector<thread> th;
    for (int g = 0; g < W; ++g) {
        th.emplace_back([&, g]() {
            o3dr_ctx* c = ctxs[(size_t)g];
            try {
                push_params(c);
                const size_t base = n_acc / (size_t)W, rem = n_acc % (size_t)W;  // contiguous blocks (dist.shard_range)
                const size_t a = (size_t)g * base + min<size_t>((size_t)g, rem), b = a + base + ((size_t)g < rem ? 1 : 0);
                if (b > a)
                    chk(o3dr_accumulate_frames_kp(c, disp.data() + a * dsz, (int64_t)dsz, cols, bgr.data() + a * csz, (int64_t)csz,
                                                  3 * (int64_t)cols, rows, cols, poses.data() + 16 * a, (int32_t)(b - a),
                                                  kp_xy.empty() ? nullptr : kp_xy.data(), kp_xy.empty() ? nullptr : kp_off.data() + a,
                                                  O3DR_MEM_HOST),
                        "accumulate_frames");
            } catch (const exception& e) {
                errors[(size_t)g] = e.what();
            }
            // (every rank must enter the collective, also after a failure of its own frames: its cloud is then empty)
            if (c) {
                const int rc = o3dr_merge_partitioned(c, comms[(size_t)g], 1, g == 0 ? cloud_small->points.data() : nullptr, g == 0 ? cap : 0,
                                                      &n_out[(size_t)g], &n_total[(size_t)g], &st[(size_t)g], O3DR_MEM_HOST);
                if (rc != O3DR_OK && errors[(size_t)g].empty()) errors[(size_t)g] = string("o3dr_merge_partitioned: ") + o3dr_last_error();
                (void)o3dr_ctx_destroy(c);
            }
        });
    }
    for (thread& t : th) t.join();
    for (int g = 0; g < W; ++g) (void)o3dr_comm_destroy(comms[(size_t)g]);
    if (reg_disp) (void)o3dr_host_unregister(disp.data());
    if (reg_bgr) (void)o3dr_host_unregister(bgr.data());
    for (int g = 0; g < W; ++g)
        if (!errors[(size_t)g].empty()) throw runtime_error("GPU " + to_string(devs[(size_t)g]) + ": " + errors[(size_t)g]);
    cloud_small->points.resize((size_t)n_out[0]);
    if (st[0] & O3DR_STATUS_VOXEL_OVERFLOW)
        cerr << "[pcl::VoxelGrid::applyFilter] Leaf size is too small for the input dataset. Integer indices would overflow." << endl;
    const double dt = chrono::duration<double>(clk::now() - t0).count();
    cout << "\n" << W << " GPU(s): cloud_big " << n_total[0] << " points -> cloud " << n_out[0] << " points, " << dt << " sec" << endl;
}

}  // namespace o3dr_host
