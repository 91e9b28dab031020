// replay_main.cpp — ROS-free replay harness for the drop-in classes (SURVEY.md 8(d) configs[0] substitute):
//   vins_replay fe <frames.bin> <out.txt>   frames.bin = int32 n, w, h, pub_every ; n * w*h bytes
// feeds the frames through FeatureTracker::readImage exactly as img_callback does (feature_tracker_node.cpp:86-111:
// readImage, then updateID for every feature) and dumps, per frame, ids / cur_pts / track_cnt / cur_un_pts / velocity.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "feature_tracker.h"

int main(int argc, char** argv) {
    if (argc < 4 || strcmp(argv[1], "fe")) { fprintf(stderr, "usage: vins_replay fe <frames.bin> <out.txt>\n"); return 2; }
    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror("frames"); return 2; }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2], pub_every = hdr[3];
    COL = w; ROW = h;
    std::vector<unsigned char> buf((size_t)w * h);
    FeatureTracker tracker;
    FILE* o = fopen(argv[3], "w");
    for (int k = 0; k < n; ++k) {
        if (fread(buf.data(), 1, buf.size(), f) != buf.size()) return 2;
        PUB_THIS_FRAME = (k % pub_every) == 0;
        cv::Mat img(h, w, cv::CV_8UC1, buf.data(), (size_t)w);
        tracker.readImage(img, 0.05 * k);
        for (unsigned int i = 0;; i++) if (!tracker.updateID(i)) break;
        fprintf(o, "frame %d n %zu\n", k, tracker.cur_pts.size());
        for (size_t i = 0; i < tracker.cur_pts.size(); ++i)
            fprintf(o, "%d %d %.9g %.9g %.9g %.9g %.9g %.9g\n", tracker.ids[i], tracker.track_cnt[i], tracker.cur_pts[i].x, tracker.cur_pts[i].y,
                    tracker.cur_un_pts[i].x, tracker.cur_un_pts[i].y, tracker.pts_velocity[i].x, tracker.pts_velocity[i].y);
    }
    fclose(o);
    fclose(f);
    return 0;
}
