// Demo caller of the two entry points, the counterpart of the reference's src/main.cpp:15-47 (which hard-codes
// its paths); here the configuration files come from the command line:
//   pwicp_demo 4d   <config_4d.txt>  <startEpoch> <epochNum> <pairMode> [overlapThd]
//   pwicp_demo pair <config_pair.txt> <output prefix>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" bool PiecewiseICP_pair_call(const char* confile, const char* outfile);
extern "C" bool PiecewiseICP_4D_call(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd);

int main(int argc, char** argv) {
    // eight hardware queues for the series' streams (the runtime's default is four; read when the runtime starts - this process is
    // ours and still single-threaded, the library itself never touches the environment)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (argc >= 6 && !strcmp(argv[1], "4d")) {
        const bool ok = PiecewiseICP_4D_call(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? (float)atof(argv[6]) : 0.75f);
        printf(ok ? "\n\n4D point cloud registration completed!\n" : "\n\n4D point cloud registration fail!!\n");
        return ok ? 0 : 1;
    }
    if (argc >= 4 && !strcmp(argv[1], "pair")) {
        const bool ok = PiecewiseICP_pair_call(argv[2], argv[3]);
        printf(ok ? "\n\nPairwise registration completed!\n" : "\n\nPairwise registration fail!!\n");
        return ok ? 0 : 1;
    }
    fprintf(stderr, "usage: %s 4d <config> <startEpoch> <epochNum> <pairMode> [overlapThd]\n       %s pair <config> <output prefix>\n", argv[0], argv[0]);
    return 2;
}
