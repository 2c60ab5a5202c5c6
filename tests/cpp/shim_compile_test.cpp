#include <vector>
#include <memory>
#include "lslam_scan_match.hpp"
struct Pt { float x,y,z,pad,intensity,p1,p2,p3; };
struct Cloud { std::vector<Pt> points; };
struct Ang { float r; float rad() const { return r; } Ang& operator=(float v){ r=v; return *this; } };
struct Tw { Ang rot_x, rot_y, rot_z; float p[3]; float& pos(int i){ return p[i]; } };
struct Iso { float m[16]; struct M { float* m; float& operator()(int r,int c){return m[r*4+c];} }; M matrix(){ return M{m}; } };
int main(){ std::shared_ptr<const Cloud> a(new Cloud), b(new Cloud); Iso T{}; for(int i=0;i<4;i++) T.m[i*5]=1;
  try { lidar_slam::ScanMatch sm(10); sm.setConvergeThreshold(0.1f,0.1f); sm.setUseCore(false); Tw tw{}; bool ok = sm.scanMatchScan(a,b,a,b,T); ok = sm.scanMatchScan(a,b,a,b,tw) || ok; ok = sm.scanMatchLocal(a,b,a,b,T) || ok; ok = sm.scanMatchLocal(a,b,a,b,tw) || ok; return ok; } catch (std::exception& e) { std::cout << "expected on CPU: " << e.what() << std::endl; return 0; } }
