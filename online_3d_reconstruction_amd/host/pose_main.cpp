// pose_main.cpp — `./pose first last --voxel_size .. --jump_pixels ..` (pose.cpp:727-767: main only
// constructs Pose inside a try/catch and prints what was thrown).
#include <iostream>

#include "o3dr_host.h"

int main(int argc, char* argv[])
{
    try {
        o3dr_host::Pose pose(argc, argv);
    } catch (const char* s) {
        std::cout << "\n" << s << std::endl;
        return 1;
    } catch (const std::string& s) {
        std::cout << "\n" << s << std::endl;
        return 1;
    } catch (const std::exception& e) {
        std::cout << "\nException: " << e.what() << std::endl;
        return 1;
    } catch (...) {
        std::cout << "\nunknown exception" << std::endl;
        return 1;
    }
    return 0;
}
