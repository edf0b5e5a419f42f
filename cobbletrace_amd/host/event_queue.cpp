// event_queue.cpp -- the key-event ring RayThread drains (eventQueue.cpp:5-24).
// Same observable behaviour: fixed-capacity ring, indices wrap when they reach the
// capacity, no overflow protection (a full ring overwrites its oldest entries).
#include "cobbletrace_host.h"

void AddEvent(eventManager_t *em, event_t event) {
    if (em->writeIndex >= em->capacity) em->writeIndex = 0;
    em->queue[em->writeIndex++] = event;
    em->active++;
}

event_t *ReadEvent(eventManager_t *em) {
    if (em->active == 0) return nullptr;
    if (em->readIndex >= em->capacity) em->readIndex = 0;
    em->active--;
    return &em->queue[em->readIndex++];
}
