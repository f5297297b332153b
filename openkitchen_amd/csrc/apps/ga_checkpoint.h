// ga_checkpoint.h -- the GA checkpoint in the reference's own text format, so that networks trained here load into the
// reference's EvolutionaryRacer and the other way round (EvolutionaryRacer/Network.hpp:29-51 writeMatrixToFile, :53-80
// readMatrixFromFile, :97,108-116 kInitFromCheckpoint; MiscUtils.hpp:52-59 agent_weights_{1,2}.txt).  Host-only, no GPU.
#pragma once

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "okenv_math.h" // the padded per-agent weight layout (OK_MLP_HID_PAD, OK_MLP_OUT, OK_MLP_OUT_PAD)

namespace ga_checkpoint
{
// genetic::writeMatrixToFile / readMatrixFromFile (EvolutionaryRacer/Network.hpp:29-51, 53-80)
// "rows cols\n", then one line per row, entries separated by one blank, printed by operator<< of a float (six significant
// digits); read back entry by entry as double and narrowed.  `m` is row-major rows x cols.
inline bool writeMatrixToFile(const std::string &filename, const std::vector<float> &m, const int rows, const int cols)
{
    std::ofstream file(filename);
    if (!file.is_open())
    {
        std::fprintf(stderr, "Unable to open file %s\n", filename.c_str());
        return false;
    }
    file << rows << " " << cols << std::endl;
    for (int i = 0; i < rows; i++)
    {
        for (int j = 0; j < cols; j++)
        {
            file << m[static_cast<size_t>(i) * cols + j];
            if (j < cols - 1)
                file << " ";
        }
        file << "\n";
    }
    return true;
}

inline bool readMatrixFromFile(const std::string &filename, std::vector<float> &m, int &rows, int &cols)
{
    std::ifstream file(filename);
    if (!file.is_open())
        return false;
    file >> rows;
    file >> cols;
    if (!file || rows < 1 || cols < 1 || rows > 4096 || cols > 4096)
        return false;
    m.assign(static_cast<size_t>(rows) * cols, 0.F);
    for (auto &e : m)
    {
        double item = 0.0;
        file >> item;
        e = static_cast<float>(item);
    }
    return static_cast<bool>(file);
}

// one agent's padded weight block (include/okenv_math.h: w1[(R+2)][32] then w2[32][8]) <-> weights_1_ ((R+2) x H), weights_2_ (H x 6)
inline void unpadWeights(const float *block, const int R, const int H, std::vector<float> &w1, std::vector<float> &w2)
{
    w1.resize(static_cast<size_t>(R + 2) * H);
    w2.resize(static_cast<size_t>(H) * OK_MLP_OUT);
    for (int i = 0; i < R + 2; ++i)
        for (int j = 0; j < H; ++j)
            w1[static_cast<size_t>(i) * H + j] = block[i * OK_MLP_HID_PAD + j];
    const float *b2 = block + (R + 2) * OK_MLP_HID_PAD;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < OK_MLP_OUT; ++j)
            w2[static_cast<size_t>(i) * OK_MLP_OUT + j] = b2[i * OK_MLP_OUT_PAD + j];
}

inline void padWeights(const std::vector<float> &w1, const std::vector<float> &w2, const int R, const int H, float *block)
{
    std::fill(block, block + OK_MLP_WEIGHTS(R), 0.F);
    for (int i = 0; i < R + 2; ++i)
        for (int j = 0; j < H; ++j)
            block[i * OK_MLP_HID_PAD + j] = w1[static_cast<size_t>(i) * H + j];
    float *b2 = block + (R + 2) * OK_MLP_HID_PAD;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < OK_MLP_OUT; ++j)
            b2[i * OK_MLP_OUT_PAD + j] = w2[static_cast<size_t>(i) * OK_MLP_OUT + j];
}
} // namespace ga_checkpoint
