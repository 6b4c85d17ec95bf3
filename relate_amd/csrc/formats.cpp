// formats.cpp
