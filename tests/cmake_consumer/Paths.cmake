# stands for the reference's Paths.cmake (:1-5): dependency paths hard-coded to a home directory that does not exist here
set(RAYLIB_CPP_INCLUDE_DIR "/home/nobody/Downloads/raylib-cpp/include")
set(RAYLIB_INCLUDE_DIR "/home/nobody/Downloads/raylib-cpp/build/_deps/raylib-build/raylib/include")
set(RAYLIB_LINK_DIR "/home/nobody/Downloads/raylib-cpp/build/_deps/raylib-build/raylib")

set(EIGEN_DIR "/usr/include/eigen3")
