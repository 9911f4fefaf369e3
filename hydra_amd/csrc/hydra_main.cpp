// placeholder until the CLI lands (next commit): keeps `make` green
#include <cstdio>
int main() { std::printf("hydra_mi355x: CLI not built yet\n"); return 2; }
