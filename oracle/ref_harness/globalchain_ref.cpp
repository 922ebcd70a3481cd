// oracle/ref_harness/globalchain_ref.cpp -- TEST INFRASTRUCTURE ONLY.
// Drives the reference's own GlobalChain<Fragment, Endpoint> (GlobalChain.h:85-189, PrioritySearchTree.h) exactly as TestGlobalChain.cpp:9-25 does:
// stdin = number of problems, then per problem the number of fragments and "xl yl xh yh" per fragment (score = xh - xl, index 0 as in the test);
// stdout = per problem: the optimal chain's fragment indices, then every fragment's final score and prev.
using namespace std;
#include "GlobalChain.h"
#include "Fragment.h"
#include <vector>
#include <iostream>
#include <cstdio>

int main() {
  int P;
  if (scanf("%d", &P) != 1) return 1;
  for (int p = 0; p < P; p++) {
    int n;
    if (scanf("%d", &n) != 1) return 1;
    vector<Fragment> fragments;
    for (int i = 0; i < n; i++) {
      int a, b, c, d;
      if (scanf("%d %d %d %d", &a, &b, &c, &d) != 4) return 1;
      fragments.push_back(Fragment(a, b, c, d, c - a, 0));
    }
    vector<Endpoint> endpoints;
    vector<int> opt;
    GlobalChain(fragments, opt, endpoints);
    printf("%d", (int)opt.size());
    for (size_t i = 0; i < opt.size(); i++) printf(" %d", opt[i]);
    printf("\n");
    for (int i = 0; i < n; i++) printf("%d %d%c", fragments[i].score, fragments[i].prev, i + 1 == n ? '\n' : ' ');
    if (n == 0) printf("\n");
  }
  return 0;
}
