// checkpoint_check.cpp -- host-only driver of openkitchen_amd/csrc/apps/ga_checkpoint.h for tests/test_ga_checkpoint.py.
//   checkpoint_check write <file> <rows> <cols> < binary float32 values on stdin
//   checkpoint_check read  <file>                > "rows cols" then binary float32 values on stdout
//   checkpoint_check pad   <R> <H>               : stdin w1 ((R+2) x H) + w2 (H x 6) as float32 -> padded block -> unpadded again on stdout
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ga_checkpoint.h"

int main(int argc, char **argv)
{
    if (argc >= 5 && !std::strcmp(argv[1], "write"))
    {
        const int          rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
        std::vector<float> m(static_cast<size_t>(rows) * cols);
        if (std::fread(m.data(), 4, m.size(), stdin) != m.size())
            return 3;
        return ga_checkpoint::writeMatrixToFile(argv[2], m, rows, cols) ? 0 : 2;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "read"))
    {
        std::vector<float> m;
        int                rows = 0, cols = 0;
        if (!ga_checkpoint::readMatrixFromFile(argv[2], m, rows, cols))
            return 2;
        std::fwrite(&rows, 4, 1, stdout);
        std::fwrite(&cols, 4, 1, stdout);
        std::fwrite(m.data(), 4, m.size(), stdout);
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "pad"))
    {
        const int          R = std::atoi(argv[2]), H = std::atoi(argv[3]);
        std::vector<float> w1(static_cast<size_t>(R + 2) * H), w2(static_cast<size_t>(H) * OK_MLP_OUT), block(OK_MLP_WEIGHTS(R), -7.F);
        if (std::fread(w1.data(), 4, w1.size(), stdin) != w1.size() || std::fread(w2.data(), 4, w2.size(), stdin) != w2.size())
            return 3;
        ga_checkpoint::padWeights(w1, w2, R, H, block.data());
        for (int i = 0; i < OK_MLP_WEIGHTS(R); ++i) // padding entries are zero, real ones are not touched by anything else
            if (!ok_mlp_weight_is_real(static_cast<uint32_t>(i), R, H) && block[i] != 0.F)
                return 4;
        std::vector<float> a, b;
        ga_checkpoint::unpadWeights(block.data(), R, H, a, b);
        std::fwrite(a.data(), 4, a.size(), stdout);
        std::fwrite(b.data(), 4, b.size(), stdout);
        return 0;
    }
    return 1;
}
