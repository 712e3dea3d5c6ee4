// placeholder until the CLI lands (see below)
int main() { return 0; }
