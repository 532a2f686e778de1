// Wrapper translation unit for the reference's CManageData.cpp (WindowToVec lives there).
#define __declspec(x)
#define _Longlong long long
#include "CManageData.cpp"
