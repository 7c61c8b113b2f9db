// main.cpp -- command line of the reference (OpticalFlowHS/main.cpp:24-148), same positional
// grammar:
//   -cl -hd  in1 in2 out alpha iterations workgroup GPU|CPU      (10 arguments)
//   -cl -cam alpha iterations workgroup GPU|CPU                  (7)
//   -cv -hd  in1 in2 out lambda iterations                       (8)
//   -cv -cam lambda iterations                                   (5)
// Images are binary PGM/PPM; the interactive getchar() pause of the original (main.cpp:87) is
// kept only when HSFLOW_PAUSE is set.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "HSOpticalFlowOpenCL.hpp"

static int runCLDisk(char *argv[])
{
    // main.cpp:91-109; alpha is parsed with atoi there, kept
    HSOpticalFlowOpenCL clOpticalFlow("OpticalFlow", argv[2], argv[3], argv[4], argv[5], (float)atoi(argv[6]),
                                      atoi(argv[7]), atoi(argv[8]), argv[9]);
    int st = clOpticalFlow.initialize();
    if (!st) st = clOpticalFlow.setup();
    if (!st) st = clOpticalFlow.run();
    clOpticalFlow.cleanup();
    return st;
}

static int runCLCamera(char *argv[])
{
    HSOpticalFlowOpenCL clOpticalFlow("OpticalFlow", argv[2], (float)atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argv[6]);
    int st = clOpticalFlow.initialize();
    if (!st) st = clOpticalFlow.setup();
    if (!st) st = clOpticalFlow.run();
    clOpticalFlow.cleanup();
    return st;
}

int main(int argc, char *argv[])
{
    int st = 0;
    if (argc == 1) {
        std::cout << "No parameters given.\n";
        return 0;
    }
    if (argc >= 3 && strcmp(argv[1], "-cl") == 0) {
        if (strcmp(argv[2], "-hd") == 0) {
            if (argc != 10) { std::cout << "Wrong argument list!\n"; return 0; }
            std::cout << "HIP disk!\n";
            st = runCLDisk(argv);
        } else if (strcmp(argv[2], "-cam") == 0) {
            if (argc != 7) { std::cout << "Wrong argument list!\n"; return 0; }
            std::cout << "HIP camera!\n";
            st = runCLCamera(argv);
        }
    } else if (argc >= 3 && strcmp(argv[1], "-cv") == 0) {
        OpticalFlowOpenCV e;
        if (strcmp(argv[2], "-hd") == 0) {
            if (argc != 8) { std::cout << "Wrong argument list!\n"; return 0; }
            std::cout << "CV-semantics disk!\n";
            st = e.runFromImg(argv[3], argv[4], argv[5], (float)atof(argv[6]), atoi(argv[7]));
        } else if (strcmp(argv[2], "-cam") == 0) {
            if (argc != 5) { std::cout << "Wrong argument list!\n"; return 0; }
            st = e.runFromCamera((float)atof(argv[3]), atoi(argv[4]));
        }
    }
    if (getenv("HSFLOW_PAUSE")) getchar();
    return st;
}
